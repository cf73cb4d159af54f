"""Training step of the speaker encoder with the GE2E loss (reference: Speaker_Embedding/Speaker_Embedding.py:27-80,
Speaker_Embedding/Modules.py:6-98) on the MI355X: dense 80 -> 256, three ZoneoutLSTM(256) with residual wrappers on the first
two, the last frame l2-normalised per row, scaled-cosine softmax loss against leave-one-out / speaker centroids with trainable
(weight, bias) = (10, -5), plain TF-Adam (eps 1e-8; the gradient-clipped op of :62-66 is overwritten at :69-72 and never runs),
learning rate max(exponential_decay, Min).  Python owns buffers and the schedule; arithmetic is libmstts_hip.so
(mstts_gemm_f32, mstts_lstm_seq_fwd/bwd, mstts_ge2e_loss_fwd_bwd, mstts_adam_tf).
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np
import torch

from . import lib
from .lib import call, gemm, ptr
from .masks import step_seed
from .params import SPK, SPK_CELL, Dims, ParamStore
from .engine import _split_k

MAX_PLANS = 3      # cached workspace sets (one per batch shape); a full-size Tacotron2 set is ~5 GB



def is_trainable(name):
    return name.startswith(SPK)


def learning_rate(step):
    """Speaker_Embedding.py:46-52."""
    from . import Hyper_Parameters as hp
    lr = hp.Speaker_Embedding.Train.Learning_Rate
    return max(lr.Initial * lr.Decay_Rate ** (step / lr.Decay_Step), lr.Min)


class SpeakerTrainEngine:
    def __init__(self, dims: Dims = None, device="cuda", seed=1234, values=None, adam=None):
        from . import Hyper_Parameters as hp
        self.d = dims or Dims()
        self.device = torch.device(device)
        self.seed = seed
        lib.load()
        self.params = ParamStore(self.d, self.device, seed=seed, values=values, trainable_fn=is_trainable, weight_reg_fn=lambda n: False)
        a = hp.Speaker_Embedding.Train.ADAM
        self.adam = adam or (a.Beta1, a.Beta2, a.Epsilon)
        # the loss's own variables 'loss/weight', 'loss/bias' (Modules.py:61-62) with their Adam slots: [w, b, pad, pad]
        self.wb = torch.tensor([10.0, -5.0, 0.0, 0.0], dtype=torch.float32, device=self.device)
        self.wb_m, self.wb_v = torch.zeros(4, device=self.device), torch.zeros(4, device=self.device)
        self.wb_mask = torch.zeros(4, dtype=torch.uint8, device=self.device)
        self.global_step = 0
        self._plans = {}          # workspace sets keyed by batch shape, least recently used first (at most MAX_PLANS kept)
        # persistent LSTM launches (csrc/persist_lstm.hip): all T steps of a layer in one launch each way; MSTTS_PERSIST_ENC=0 keeps the
        # launch-per-step loops, which are also the fallback
        self.persist_lstm = os.environ.get("MSTTS_PERSIST_ENC", "1") != "0"
        self.persist_lstm_fallbacks = 0

    def _f(self, *shape):
        n = int(np.prod(shape))
        return torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=self.device)[:n].view(shape)

    def P(self, name):
        return self.params.p(name)

    def G(self, name):
        return self.params.g(name)

    def plan(self, N, T):
        if (N, T) in self._plans:
            self._plans[(N, T)] = self._plans.pop((N, T))          # most recently used last
            return self._plans[(N, T)]
        while len(self._plans) >= MAX_PLANS:                 # variable-length training: do not keep a workspace per shape forever
            self._plans.pop(next(iter(self._plans)))
        d, f = self.d, self._f
        H, L = d.spk_lstm, d.spk_lstm_n

        class W:
            pass
        w = W()
        w.N, w.T = N, T
        w.x = [f(N, T, d.spk) for _ in range(L + 1)]            # x[0] = dense output, x[l+1] = output of cell l
        w.xw = f(N * T, 4 * H)
        w.c = [f(T + 1, N, H) for _ in range(L)]; w.h = [f(T + 1, N, H) for _ in range(L)]
        w.acts = [f(T, N, 4 * H) for _ in range(L)]; w.craw = [f(T, N, H) for _ in range(L)]
        w.zc = [torch.zeros(T * N * H, dtype=torch.uint8, device=self.device) for _ in range(L)]
        w.zh = [torch.zeros(T * N * H, dtype=torch.uint8, device=self.device) for _ in range(L)]
        lb = lib.load()
        w.gates = f(int(lb.mstts_lstm_seq_ws_floats(N, H, 0))); w.bwd_ws = f(int(lb.mstts_lstm_seq_ws_floats(N, H, 1)))
        # fused cell steps (one launch per step): packed recurrent kernel + packed h blocks; the residual wrapper's sum (output = cell
        # output + input, state untouched) is then applied once per sequence
        w.fused = bool(lb.mstts_cell_fwd_supported(H, H)) and d.spk == H
        if w.fused:
            w.whp, w.hp, w.y = f(H * 4 * H), f(2 * int(lb.mstts_cell_act_floats(N, H))), f(N, T, H)
        w.lengths = torch.full((N,), T, dtype=torch.int32, device=self.device)
        # (the persistent launches take no residual input: they need the fused form's "one add per sequence afterwards")
        w.persist = self.persist_lstm and w.fused and bool(lb.mstts_persist_lstm_supported_n(N, H, 1))
        if w.persist:
            n = int(lb.mstts_persist_lstm_pack_floats())
            w.pk = [(f(n), f(n)) for _ in range(L)]                                   # (forward order, BPTT order) per layer, refreshed per step
            w.pxch = f(int(lb.mstts_persist_lstm_ws_bytes_n(N, 1)) // 4)
            w.pctrl = torch.zeros(16, dtype=torch.int32, device=self.device)
            w.pctrl_host = torch.zeros(16, dtype=torch.int32).pin_memory()
            w.phist = [f(int(lb.mstts_persist_lstm_hist_floats_n(T, N, 1))) for _ in range(L)]   # packed history per layer (the BPTT reads it)
            w.pbws = f(int(lb.mstts_persist_lstm_bwd_floats_n(T, N, 1)))
            w.phist_valid = [False] * L
            w.groups = (N + 31) // 32
        w.loss_ws = f(int(lb.mstts_ge2e_ws_floats(N, d.spk, 256)))
        w.out3 = f(4)
        w.d_out = f(N, T, d.spk)                                # gradient of a cell's output sequence
        w.d_in = f(N, T, d.spk)
        w.dgs, w.dgp = f(T, N, 4 * H), f(N, T, 4 * H)
        self._plans[(N, T)] = w
        return w

    def forward(self, mel, w, seed=None, masks=None):
        """mel [N,T,n_mel] -> w.x[-1] [N,T,spk] (training mode: Philox or injected zoneout masks)."""
        d = self.d
        N, T, H = w.N, w.T, d.spk_lstm
        w.mel = mel
        for i in range(d.spk_lstm_n):
            if masks is not None:
                w.zc[i].copy_(torch.as_tensor(np.asarray(masks["s_zc_%d" % i], np.uint8)).reshape(-1))
                w.zh[i].copy_(torch.as_tensor(np.asarray(masks["s_zh_%d" % i], np.uint8)).reshape(-1))
            else:
                sd = seed if seed is not None else step_seed(self.seed, self.global_step)
                call("mstts_philox_keep_mask", ptr(w.zc[i]), T * N * H, sd, 60 + 2 * i, 1 - d.zoneout)
                call("mstts_philox_keep_mask", ptr(w.zh[i]), T * N * H, sd, 61 + 2 * i, 1 - d.zoneout)
        k, ok = self.P(SPK + "dense/kernel"); b, ob = self.P(SPK + "dense/bias")
        gemm(mel, k, w.x[0], N * T, d.spk, d.n_mel, d.n_mel, d.spk, d.spk, bias=b, b_off=ok, bias_off=ob)
        for i in range(d.spk_lstm_n):
            k, ok = self.P(SPK_CELL % (i, i) + "kernel"); b, ob = self.P(SPK_CELL % (i, i) + "bias")
            gemm(w.x[i], k, w.xw, N * T, 4 * H, d.spk, d.spk, 4 * H, 4 * H, bias=b, b_off=ok, bias_off=ob)
            q = lib.LstmSeqFwd()
            q.B, q.T, q.H = N, T, H
            q.xw = ptr(w.xw); q.wh = ptr(k, ok + d.spk * 4 * H); q.wh_ld = 4 * H
            q.lengths = ptr(w.lengths); q.reverse = 0; q.zoneout = d.zoneout
            q.zc, q.zh = ptr(w.zc[i]), ptr(w.zh[i])
            res = i < d.spk_lstm_n - 1
            q.residual = ptr(w.x[i]) if (res and not w.fused) else None
            q.out = ptr(w.y if (res and w.fused) else w.x[i + 1]); q.out_sb = T * H; q.out_st = H
            q.c_hist, q.h_hist, q.acts, q.c_raw = ptr(w.c[i]), ptr(w.h[i]), ptr(w.acts[i]), ptr(w.craw[i])
            q.gates_ws = ptr(w.gates)
            done = False
            if getattr(w, "persist", False):
                call("mstts_persist_lstm_pack", ptr(k, ok + d.spk * 4 * H), 4 * H, ptr(w.pk[i][0]), ptr(w.pk[i][1]))
                call("mstts_lstm_seq_fwd_persistent", C.byref(q), ptr(w.pk[i][0]), ptr(w.pxch), ptr(w.pctrl), ptr(w.phist[i]))
                done = w.phist_valid[i] = self._persist_ok(w, 32 * w.groups)
            if not done:
                if w.fused:
                    call("mstts_pack_cell_fwd", ptr(k, ok + d.spk * 4 * H), 4 * H, ptr(w.whp), H, H)
                    q.wh_p, q.h_p = ptr(w.whp), ptr(w.hp)
                call("mstts_lstm_seq_fwd", C.byref(q))
            if res and w.fused:
                call("mstts_add", ptr(w.y), ptr(w.x[i]), ptr(w.x[i + 1]), N * T * H)
        return w.x[-1]

    def _persist_ok(self, w, n_wg):
        """Control words of the persistent launch just enqueued: True when it ran to its end (else the caller runs the launch-per-step loop)."""
        w.pctrl_host.copy_(w.pctrl, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        ok = int(w.pctrl_host[1]) == 0 and int(w.pctrl_host[2]) == n_wg
        if not ok:
            self.persist_lstm_fallbacks += 1
        return ok

    def loss_and_backward(self, w, batch_per_speaker):
        d = self.d
        N, T, H, Dm = w.N, w.T, d.spk_lstm, d.spk
        P_ = batch_per_speaker
        S = N // P_
        assert S * P_ == N
        self.params.grad.zero_()
        w.d_out.zero_()
        # loss on the last frame; its gradient lands in the last time row of d_out
        last = (T - 1) * Dm
        call("mstts_ge2e_loss_fwd_bwd", ptr(w.x[-1], last), T * Dm, S, P_, Dm, ptr(self.wb), ptr(w.out3), ptr(w.d_out, last), T * Dm, ptr(w.loss_ws))
        d_out = w.d_out
        for i in range(d.spk_lstm_n - 1, -1, -1):
            k, ok = self.P(SPK_CELL % (i, i) + "kernel")
            q = lib.LstmSeqBwd()
            q.B, q.T, q.H = N, T, H
            q.wh = ptr(k, ok + Dm * 4 * H); q.wh_ld = 4 * H
            q.lengths = ptr(w.lengths); q.reverse = 0; q.zoneout = d.zoneout
            q.zc, q.zh = ptr(w.zc[i]), ptr(w.zh[i])
            q.d_out = ptr(d_out); q.dout_sb = T * H; q.dout_st = H
            q.c_hist, q.acts, q.c_raw = ptr(w.c[i]), ptr(w.acts[i]), ptr(w.craw[i])
            q.dgates_step, q.dgates_pos, q.ws = ptr(w.dgs), ptr(w.dgp), ptr(w.bwd_ws)
            done = False
            if getattr(w, "persist", False) and w.phist_valid[i]:       # (the persistent BPTT reads the packed history of a persistent forward)
                call("mstts_lstm_seq_bwd_persistent", C.byref(q), ptr(w.pk[i][1]), ptr(w.pxch), ptr(w.pctrl), ptr(w.phist[i]), ptr(w.pbws))
                done = self._persist_ok(w, 16 * w.groups)
            if not done:
                call("mstts_lstm_seq_bwd", C.byref(q))
            gk, ogk = self.G(SPK_CELL % (i, i) + "kernel"); gb, ogb = self.G(SPK_CELL % (i, i) + "bias")
            gemm(w.x[i], w.dgp, gk, Dm, 4 * H, N * T, Dm, 4 * H, 4 * H, trans_a=True, split_k=max(2, _split_k(Dm, 4 * H, N * T)), c_off=ogk)
            gemm(w.h[i], w.dgs, gk, H, 4 * H, N * T, H, 4 * H, 4 * H, trans_a=True, split_k=max(2, _split_k(H, 4 * H, N * T)), c_off=ogk + Dm * 4 * H)
            call("mstts_colsum", ptr(w.dgs), N * T, 4 * H, 4 * H, ptr(gb, ogb), 1)
            d_in = w.d_in if d_out is w.d_out else w.d_out
            gemm(w.dgp, k, d_in, N * T, Dm, 4 * H, 4 * H, 4 * H, Dm, trans_b=True, b_off=ok)
            if i < d.spk_lstm_n - 1:                             # ResidualWrapper: the output gradient also reaches the input
                call("mstts_add", ptr(d_in), ptr(d_out), ptr(d_in), N * T * Dm)
            d_out = d_in
        gk, ogk = self.G(SPK + "dense/kernel"); gb, ogb = self.G(SPK + "dense/bias")
        gemm(w.mel, d_out, gk, d.n_mel, Dm, N * T, d.n_mel, Dm, Dm, trans_a=True, split_k=max(2, _split_k(d.n_mel, Dm, N * T)), c_off=ogk)
        call("mstts_colsum", ptr(d_out), N * T, Dm, Dm, ptr(gb, ogb), 1)

    def adam_step(self, w):
        ps = self.params
        b1, b2, eps = self.adam
        t = self.global_step + 1
        lr = learning_rate(self.global_step)
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        call("mstts_adam_tf", ptr(ps.train), ptr(ps.grad), ptr(ps.adam_m), ptr(ps.adam_v), ptr(ps.wd_mask), 0.0, 1.0, float(lr_t), b1, b2, eps, ps.n_train)
        # the loss's (weight, bias): gradients sit in out3[1:3]
        call("mstts_adam_tf", ptr(self.wb), ptr(w.out3, 1), ptr(self.wb_m), ptr(self.wb_v), ptr(self.wb_mask), 0.0, 1.0, float(lr_t), b1, b2, eps, 2)
        self.global_step += 1
        ps.touch()                           # (an InferEngine sharing this store keys its packed recurrent kernels on the version)
        return lr

    def train_step(self, mel, batch_per_speaker, masks=None, seed=None):
        N, T, _ = mel.shape
        w = self.plan(N, T)
        self.forward(mel, w, seed=seed, masks=masks)
        self.loss_and_backward(w, batch_per_speaker)
        self.adam_step(w)
        return w
