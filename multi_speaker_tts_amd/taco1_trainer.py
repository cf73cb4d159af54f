"""Training step of the Taco1 mel -> spectrogram vocoder (reference: Taco1_Mel_to_Spect/Taco1_Mel_to_Spect.py:24-100 with
Taco1_Mel_to_Spect/Modules.py:8-108) on the MI355X: ConvBank (8 convs + BN, max-pool, two projections + BN, residual) ->
4 highway layers -> BiLSTM(128, zoneout .1, no length mask) -> dense 256 -> 1025; loss = mean |pred - spectrogram| +
1e-6 * sum l2_loss(v) over variables whose lower-cased name has none of 'bias', 'lstm', 'rnn'; TF-Adam; BN moving statistics
updated by the same step.  Python owns buffers and the schedule, the arithmetic is libmstts_hip.so (the same GEMM / BN /
LSTM-sequence kernels as the Tacotron2 trainer plus mstts_maxpool2_same_bwd, mstts_highway_combine_bwd, mstts_l1_loss_fwd_bwd).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import lib
from .lib import ACT_NONE, ACT_RELU, call, gemm, ptr
from .masks import step_seed
from .params import VOC, Dims, ParamStore, bank_suffix
from .engine import BN_EPS, BN_MOM, _split_k

BIRNN = VOC + "birnn/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/"

MAX_PLANS = 3      # cached workspace sets (one per batch shape); a full-size Tacotron2 set is ~5 GB



def is_trainable(name):
    return name.startswith(VOC) and not name.endswith(("moving_mean", "moving_variance"))


def in_weight_reg(name):
    """Taco1_Mel_to_Spect.py:45-53."""
    low = name.lower()
    return not any(s in low for s in ("bias", "lstm", "rnn"))


def learning_rate(step):
    """Taco1_Mel_to_Spect.py:63-69."""
    from . import Hyper_Parameters as hp
    lr = hp.Taco1_Mel_to_Spect.Train.Learning_Rate
    v = lr.Initial * lr.Decay_Rate ** ((step - lr.Decay_Start_Step) / lr.Decay_Step)
    return min(max(v, lr.Min), lr.Initial)


class Taco1TrainEngine:
    def __init__(self, dims: Dims = None, device="cuda", seed=1234, values=None, wr_rate=None, adam=None):
        from . import Hyper_Parameters as hp
        self.d = dims or Dims()
        self.device = torch.device(device)
        self.seed = seed
        lib.load()
        self.params = ParamStore(self.d, self.device, seed=seed, values=values, trainable_fn=is_trainable, weight_reg_fn=in_weight_reg)
        tr = hp.Taco1_Mel_to_Spect.Train
        self.wr_rate = tr.Weight_Regularization_Rate if wr_rate is None else wr_rate
        self.adam = adam or (tr.ADAM.Beta1, tr.ADAM.Beta2, tr.ADAM.Epsilon)
        self.global_step = 0
        self.flip = {}
        self._plans = {}          # workspace sets keyed by batch shape, least recently used first (at most MAX_PLANS kept)

    def _f(self, *shape):
        n = int(np.prod(shape))
        return torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=self.device)[:n].view(shape)

    def P(self, name):
        return self.params.p(name)

    def G(self, name):
        return self.params.g(name)

    # ------------------------------------------------------------------ buffers
    def plan(self, B, S):
        if (B, S) in self._plans:
            self._plans[(B, S)] = self._plans.pop((B, S))          # most recently used last
            return self._plans[(B, S)]
        while len(self._plans) >= MAX_PLANS:                 # variable-length training: do not keep a workspace per shape forever
            self._plans.pop(next(iter(self._plans)))
        d, f = self.d, self._f

        class W:
            pass
        w = W()
        w.B, w.S = B, S
        rows, C1, Hh = B * S, d.bank_k * d.bank_ch, d.birnn
        w.bank_a = [f(rows, d.bank_ch) for _ in range(d.bank_k)]
        w.bank_mean = [f(d.bank_ch) for _ in range(d.bank_k)]; w.bank_rstd = [f(d.bank_ch) for _ in range(d.bank_k)]
        w.cat, w.pool = f(rows, C1), f(rows, C1)
        w.p1_a, w.p1_y, w.p1_mean, w.p1_rstd = f(rows, d.proj1_ch), f(rows, d.proj1_ch), f(d.proj1_ch), f(d.proj1_ch)
        w.p2_a, w.p2_y, w.p2_mean, w.p2_rstd = f(rows, d.n_mel), f(rows, d.n_mel), f(d.n_mel), f(d.n_mel)
        w.hx = [f(rows, d.n_mel) for _ in range(d.highway_n + 1)]
        w.hh = [f(rows, d.n_mel) for _ in range(d.highway_n)]; w.ht = [f(rows, d.n_mel) for _ in range(d.highway_n)]
        w.xw = {dr: f(rows, 4 * Hh) for dr in ("fw", "bw")}
        w.c = {dr: f(S + 1, B, Hh) for dr in ("fw", "bw")}; w.h = {dr: f(S + 1, B, Hh) for dr in ("fw", "bw")}
        w.acts = {dr: f(S, B, 4 * Hh) for dr in ("fw", "bw")}; w.craw = {dr: f(S, B, Hh) for dr in ("fw", "bw")}
        w.zc = {dr: torch.zeros(S * B * Hh, dtype=torch.uint8, device=self.device) for dr in ("fw", "bw")}
        w.zh = {dr: torch.zeros(S * B * Hh, dtype=torch.uint8, device=self.device) for dr in ("fw", "bw")}
        lb = lib.load()
        w.gates = f(int(lb.mstts_lstm_seq_ws_floats(B, Hh, 0)))
        w.bwd_ws = {dr: f(int(lb.mstts_lstm_seq_ws_floats(B, Hh, 1))) for dr in ("fw", "bw")}
        # fused cell steps (one launch per step for both directions): packed recurrent kernels + packed h blocks
        w.fused = bool(lb.mstts_cell_fwd_supported(Hh, Hh))
        if w.fused:
            w.whp = {dr: f(Hh * 4 * Hh) for dr in ("fw", "bw")}
            w.hp = {dr: f(2 * int(lb.mstts_cell_act_floats(B, Hh))) for dr in ("fw", "bw")}
        w.rnn = f(rows, 2 * Hh)
        w.pred, w.d_pred = f(rows, d.n_spec), f(rows, d.n_spec)
        w.lengths = torch.full((B,), S, dtype=torch.int32, device=self.device)
        w.bn_ws = f(4 * max(C1, d.proj1_ch, d.n_mel, d.bank_ch) + 64)
        w.scalars = f(4)
        # backward
        w.d_rnn = f(rows, 2 * Hh)
        w.dgs = {dr: f(S, B, 4 * Hh) for dr in ("fw", "bw")}; w.dgp = {dr: f(B, S, 4 * Hh) for dr in ("fw", "bw")}
        w.dx = [f(rows, d.n_mel) for _ in range(2)]
        w.dh, w.dt = f(rows, d.n_mel), f(rows, d.n_mel)
        w.d_p1y, w.d_pool, w.d_cat = f(rows, d.proj1_ch), f(rows, C1), f(rows, C1)
        w.dz_big = f(rows, max(d.proj1_ch, d.bank_ch, d.n_mel))
        w.d_bank, w.d_mel = f(rows, d.bank_ch), f(rows, d.n_mel)
        self._plans[(B, S)] = w
        return w

    def _bn_fwd(self, prefix, a, y, mean, rstd, rows, Cc, ws):
        g, og = self.P(prefix + "gamma"); b, ob = self.P(prefix + "beta")
        mm, omm = self.P(prefix + "moving_mean"); mv, omv = self.P(prefix + "moving_variance")
        call("mstts_bn_train_fwd", ptr(a), ptr(g, og), ptr(b, ob), ptr(mm, omm), ptr(mv, omv), ptr(y), ptr(mean), ptr(rstd),
             None, 1.0, BN_MOM, BN_EPS, rows, Cc, ptr(ws))

    def _conv(self, x, rows, T, cin, cout, K, name, out, act, lda=None):
        k, ok = self.P(name + "/kernel"); b, ob = self.P(name + "/bias")
        gemm(x, k, out, rows, cout, K * cin, cin, cout, cout, bias=b, act=act, win=(T, cin, (K - 1) // 2), b_off=ok, bias_off=ob)

    def _conv_bn_bwd(self, dy, x_in, a, mean, rstd, act, conv, bn, rows, T, cin, cout, K, dz, dx, dx_accumulate=False):
        """y = BN(act(conv(x))): dy -> dz (conv pre-activation grad), parameter grads, dx (+= when dx_accumulate)."""
        g, og = self.P(bn + "gamma"); gg, ogg = self.G(bn + "gamma"); gb, ogb = self.G(bn + "beta"); gbias, ogbias = self.G(conv + "/bias")
        call("mstts_bn_train_bwd", ptr(dy), ptr(a), ptr(g, og), ptr(mean), ptr(rstd), None, 1.0, act, ptr(dz),
             ptr(gg, ogg), ptr(gb, ogb), ptr(gbias, ogbias), rows, cout, ptr(self._w.bn_ws))
        gk, ogk = self.G(conv + "/kernel")
        pad = (K - 1) // 2
        gemm(x_in, dz, gk, K * cin, cout, rows, cin, cout, cout, trans_a=True, win=(T, cin, pad),
             split_k=max(2, _split_k(K * cin, cout, rows)), c_off=ogk)
        if dx is not None:
            k, ok = self.P(conv + "/kernel")
            key = (conv, K, cin, cout)
            if key not in self.flip:
                self.flip[key] = self._f(K, cout, cin)
            wt = self.flip[key]
            call("mstts_conv_kernel_flip", ptr(k, ok), ptr(wt), K, cin, cout)
            gemm(dz, wt, dx, rows, cin, K * cout, cout, cin, cin, win=(T, cout, K - 1 - pad), accumulate=dx_accumulate)

    # ------------------------------------------------------------------ forward
    def forward(self, mel, w, seed=None, masks=None):
        """mel [B,S,n_mel] device tensor -> w.pred [B*S, n_spec]; saves what the backward needs."""
        d = self.d
        B, S = w.B, w.S
        rows, C1, Hh = B * S, d.bank_k * d.bank_ch, d.birnn
        self._w = w
        w.mel = mel
        if masks is not None:
            for dr in ("fw", "bw"):
                w.zc[dr].copy_(torch.as_tensor(np.asarray(masks["v_zc_" + dr], np.uint8)).reshape(-1))
                w.zh[dr].copy_(torch.as_tensor(np.asarray(masks["v_zh_" + dr], np.uint8)).reshape(-1))
        else:
            sd = seed if seed is not None else step_seed(self.seed, self.global_step)
            for i, dr in enumerate(("fw", "bw")):
                call("mstts_philox_keep_mask", ptr(w.zc[dr]), S * B * Hh, sd, 50 + 2 * i, 1 - d.zoneout)
                call("mstts_philox_keep_mask", ptr(w.zh[dr]), S * B * Hh, sd, 51 + 2 * i, 1 - d.zoneout)
        for k in range(1, d.bank_k + 1):
            sfx = bank_suffix(k)
            self._conv(mel, rows, S, d.n_mel, d.bank_ch, k, VOC + "convbank_0/conv1d%s" % sfx, w.bank_a[k - 1], ACT_RELU)
            y = w.dz_big                                   # scratch for the normalised block before it is copied into the concat
            self._bn_fwd(VOC + "convbank_0/batch_normalization%s/" % sfx, w.bank_a[k - 1], y, w.bank_mean[k - 1], w.bank_rstd[k - 1], rows, d.bank_ch, w.bn_ws)
            call("mstts_copy2d", ptr(y), d.bank_ch, ptr(w.cat, (k - 1) * d.bank_ch), C1, rows, d.bank_ch, 0)
        call("mstts_maxpool2_same", ptr(w.cat), ptr(w.pool), B, S, C1)
        self._conv(w.pool, rows, S, C1, d.proj1_ch, d.proj1_k, VOC + "convbank_0/conv1d_8", w.p1_a, ACT_RELU)
        self._bn_fwd(VOC + "convbank_0/batch_normalization_8/", w.p1_a, w.p1_y, w.p1_mean, w.p1_rstd, rows, d.proj1_ch, w.bn_ws)
        self._conv(w.p1_y, rows, S, d.proj1_ch, d.n_mel, d.proj2_k, VOC + "convbank_0/conv1d_9", w.p2_a, ACT_NONE)
        self._bn_fwd(VOC + "convbank_0/batch_normalization_9/", w.p2_a, w.p2_y, w.p2_mean, w.p2_rstd, rows, d.n_mel, w.bn_ws)
        call("mstts_add", ptr(mel), ptr(w.p2_y), ptr(w.hx[0]), rows * d.n_mel)
        for i in range(d.highway_n):
            pre = VOC + "highway_%d/" % i
            for nm, out in (("dense", w.hh[i]), ("dense_1", w.ht[i])):
                kk, ok = self.P(pre + nm + "/kernel"); b, ob = self.P(pre + nm + "/bias")
                gemm(w.hx[i], kk, out, rows, d.n_mel, d.n_mel, d.n_mel, d.n_mel, d.n_mel, bias=b, b_off=ok, bias_off=ob)
            call("mstts_highway_combine", ptr(w.hh[i]), ptr(w.ht[i]), ptr(w.hx[i]), ptr(w.hx[i + 1]), rows * d.n_mel)
        x = w.hx[d.highway_n]
        seqs = []
        for di, dr in enumerate(("fw", "bw")):
            k, ok = self.P(BIRNN % dr + "kernel"); b, ob = self.P(BIRNN % dr + "bias")
            gemm(x, k, w.xw[dr], rows, 4 * Hh, d.n_mel, d.n_mel, 4 * Hh, 4 * Hh, bias=b, b_off=ok, bias_off=ob)
            q = lib.LstmSeqFwd()
            q.B, q.T, q.H = B, S, Hh
            q.xw = ptr(w.xw[dr]); q.wh = ptr(k, ok + d.n_mel * 4 * Hh); q.wh_ld = 4 * Hh
            q.lengths = ptr(w.lengths); q.reverse = di; q.zoneout = d.zoneout
            q.zc, q.zh = ptr(w.zc[dr]), ptr(w.zh[dr])
            q.out = ptr(w.rnn, di * Hh); q.out_sb = S * 2 * Hh; q.out_st = 2 * Hh
            q.c_hist, q.h_hist, q.acts, q.c_raw = ptr(w.c[dr]), ptr(w.h[dr]), ptr(w.acts[dr]), ptr(w.craw[dr])
            q.gates_ws = ptr(w.gates)
            if w.fused:
                call("mstts_pack_cell_fwd", ptr(k, ok + d.n_mel * 4 * Hh), 4 * Hh, ptr(w.whp[dr]), Hh, Hh)
                q.wh_p, q.h_p = ptr(w.whp[dr]), ptr(w.hp[dr])
            seqs.append(q)
        call("mstts_lstm_seq_fwd_pair", C.byref(seqs[0]), C.byref(seqs[1]))     # both directions advance together
        kk, ok = self.P(VOC + "dense/kernel"); b, ob = self.P(VOC + "dense/bias")
        gemm(w.rnn, kk, w.pred, rows, d.n_spec, 2 * Hh, 2 * Hh, d.n_spec, d.n_spec, bias=b, b_off=ok, bias_off=ob)
        return w.pred

    # ------------------------------------------------------------------ loss + backward
    def loss_and_backward(self, w, spectrogram):
        d = self.d
        B, S = w.B, w.S
        rows, C1, Hh = B * S, d.bank_k * d.bank_ch, d.birnn
        ps = self.params
        ps.grad.zero_()
        w.scalars.zero_()
        call("mstts_l1_loss_fwd_bwd", ptr(w.pred), ptr(spectrogram), rows * d.n_spec, ptr(w.scalars), ptr(w.d_pred))
        call("mstts_l2_loss_acc", ptr(ps.train), ptr(ps.wd_mask), ps.n_train, ptr(w.scalars, 1))
        # projection
        kk, ok = self.P(VOC + "dense/kernel"); gk, ogk = self.G(VOC + "dense/kernel"); gb, ogb = self.G(VOC + "dense/bias")
        gemm(w.rnn, w.d_pred, gk, 2 * Hh, d.n_spec, rows, 2 * Hh, d.n_spec, d.n_spec, trans_a=True, split_k=max(2, _split_k(2 * Hh, d.n_spec, rows)), c_off=ogk)
        call("mstts_colsum", ptr(w.d_pred), rows, d.n_spec, d.n_spec, ptr(gb, ogb), 1)
        gemm(w.d_pred, kk, w.d_rnn, rows, 2 * Hh, d.n_spec, d.n_spec, d.n_spec, 2 * Hh, trans_b=True, b_off=ok)
        # BiLSTM
        x_in = w.hx[d.highway_n]
        dy = w.dx[0]
        bseqs = []
        for di, dr in enumerate(("fw", "bw")):
            k, ok = self.P(BIRNN % dr + "kernel")
            q = lib.LstmSeqBwd()
            q.B, q.T, q.H = B, S, Hh
            q.wh = ptr(k, ok + d.n_mel * 4 * Hh); q.wh_ld = 4 * Hh
            q.lengths = ptr(w.lengths); q.reverse = di; q.zoneout = d.zoneout
            q.zc, q.zh = ptr(w.zc[dr]), ptr(w.zh[dr])
            q.d_out = ptr(w.d_rnn, di * Hh); q.dout_sb = S * 2 * Hh; q.dout_st = 2 * Hh
            q.c_hist, q.acts, q.c_raw = ptr(w.c[dr]), ptr(w.acts[dr]), ptr(w.craw[dr])
            q.dgates_step, q.dgates_pos, q.ws = ptr(w.dgs[dr]), ptr(w.dgp[dr]), ptr(w.bwd_ws[dr])
            bseqs.append(q)
        call("mstts_lstm_seq_bwd_pair", C.byref(bseqs[0]), C.byref(bseqs[1]))
        for di, dr in enumerate(("fw", "bw")):
            k, ok = self.P(BIRNN % dr + "kernel")
            gk, ogk = self.G(BIRNN % dr + "kernel"); gb, ogb = self.G(BIRNN % dr + "bias")
            gemm(x_in, w.dgp[dr], gk, d.n_mel, 4 * Hh, rows, d.n_mel, 4 * Hh, 4 * Hh, trans_a=True,
                 split_k=max(2, _split_k(d.n_mel, 4 * Hh, rows)), c_off=ogk)
            gemm(w.h[dr], w.dgs[dr], gk, Hh, 4 * Hh, rows, Hh, 4 * Hh, 4 * Hh, trans_a=True,
                 split_k=max(2, _split_k(Hh, 4 * Hh, rows)), c_off=ogk + d.n_mel * 4 * Hh)
            call("mstts_colsum", ptr(w.dgs[dr]), rows, 4 * Hh, 4 * Hh, ptr(gb, ogb), 1)
            gemm(w.dgp[dr], k, dy, rows, d.n_mel, 4 * Hh, 4 * Hh, 4 * Hh, d.n_mel, trans_b=True, accumulate=(di == 1), b_off=ok)
        # highway (reverse)
        for i in range(d.highway_n - 1, -1, -1):
            pre = VOC + "highway_%d/" % i
            dxn = w.dx[1] if dy is w.dx[0] else w.dx[0]
            call("mstts_highway_combine_bwd", ptr(w.hh[i]), ptr(w.ht[i]), ptr(w.hx[i]), ptr(dy), ptr(w.dh), ptr(w.dt), ptr(dxn), rows * d.n_mel)
            for nm, dpre in (("dense", w.dh), ("dense_1", w.dt)):
                kk, ok = self.P(pre + nm + "/kernel"); gk, ogk = self.G(pre + nm + "/kernel"); gb, ogb = self.G(pre + nm + "/bias")
                gemm(w.hx[i], dpre, gk, d.n_mel, d.n_mel, rows, d.n_mel, d.n_mel, d.n_mel, trans_a=True,
                     split_k=max(2, _split_k(d.n_mel, d.n_mel, rows)), c_off=ogk)
                call("mstts_colsum", ptr(dpre), rows, d.n_mel, d.n_mel, ptr(gb, ogb), 1)
                gemm(dpre, kk, dxn, rows, d.n_mel, d.n_mel, d.n_mel, d.n_mel, d.n_mel, trans_b=True, accumulate=True, b_off=ok)
            dy = dxn
        # conv bank (the residual `inputs + new` sends dy to the mel input too; the mel is data here, so only the bank path matters)
        self._conv_bn_bwd(dy, w.p1_y, w.p2_a, w.p2_mean, w.p2_rstd, ACT_NONE, VOC + "convbank_0/conv1d_9", VOC + "convbank_0/batch_normalization_9/",
                          rows, S, d.proj1_ch, d.n_mel, d.proj2_k, w.dz_big, w.d_p1y)
        self._conv_bn_bwd(w.d_p1y, w.pool, w.p1_a, w.p1_mean, w.p1_rstd, ACT_RELU, VOC + "convbank_0/conv1d_8", VOC + "convbank_0/batch_normalization_8/",
                          rows, S, C1, d.proj1_ch, d.proj1_k, w.dz_big, w.d_pool)
        call("mstts_maxpool2_same_bwd", ptr(w.cat), ptr(w.d_pool), ptr(w.d_cat), B, S, C1)
        for k in range(1, d.bank_k + 1):
            sfx = bank_suffix(k)
            call("mstts_copy2d", ptr(w.d_cat, (k - 1) * d.bank_ch), C1, ptr(w.d_bank), d.bank_ch, rows, d.bank_ch, 0)
            self._conv_bn_bwd(w.d_bank, w.mel, w.bank_a[k - 1], w.bank_mean[k - 1], w.bank_rstd[k - 1], ACT_RELU, VOC + "convbank_0/conv1d%s" % sfx,
                              VOC + "convbank_0/batch_normalization%s/" % sfx, rows, S, d.n_mel, d.bank_ch, k, w.dz_big, None)

    def adam_step(self):
        ps = self.params
        b1, b2, eps = self.adam
        t = self.global_step + 1
        lr = learning_rate(self.global_step)
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        call("mstts_adam_tf", ptr(ps.train), ptr(ps.grad), ptr(ps.adam_m), ptr(ps.adam_v), ptr(ps.wd_mask), float(self.wr_rate),
             1.0, float(lr_t), b1, b2, eps, ps.n_train)
        self.global_step += 1
        ps.touch()                           # (an InferEngine sharing this store keys its packed / folded kernels on the version)
        return lr

    def scalars(self, w):
        s = w.scalars.detach().cpu().numpy()
        wr = float(s[1]) * self.wr_rate
        return {"Loss": float(s[0]) + wr, "L1_Loss": float(s[0]), "Weight_Regularization_Loss": wr}

    def train_step(self, mel, spectrogram, masks=None, seed=None):
        """mel [B,S,n_mel], spectrogram [B,S,n_spec] (device tensors, contiguous)."""
        B, S, _ = mel.shape
        w = self.plan(B, S)
        self.forward(mel, w, seed=seed, masks=masks)
        self.loss_and_backward(w, spectrogram)
        self.adam_step()
        return w
