"""Inference forward of the Tacotron2 graph (Tacotron2.Inference_Mel_to_Spectrogram's Session.run,
MSTTS_SV.py:301-308) as a schedule of libmstts_hip.so calls:

  speaker encoder (Speaker_Embedding/Modules.py:6-37,127-137) -> encoder in inference mode
  (Modules.py:15-73: BN moving statistics, no dropout, deterministic zoneout) -> memory/keys ->
  free-running attention decoder with stop-token gating (Modules.py:212-237; native loop driver) ->
  postnet + residual (Modules.py:121-143) -> Taco1 mel->spectrogram (Taco1_Mel_to_Spect/Modules.py:8-105).

Prenet dropout stays active (quirk Q9): its keep-masks are Philox streams, or injected for tests.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import warnings

import numpy as np
import torch

from . import lib
from .lib import ACT_NONE, ACT_RELU, ACT_TANH, gemm, ptr, call
from .masks import MaskSet, step_seed
from .params import CELL, ENC_CELL, LSA, SPK, SPK_CELL, VOC, VOC_CELL, Dims, ParamStore, bank_suffix

BN_EPS = 1e-3
SPK_OVERLAP = os.environ.get("MSTTS_SPK_OVERLAP", "1") != "0"     # inference forward: the speaker stack on its own stream beside the text encoder


class _PersistRetry(RuntimeError):
    pass


def to_host(tensors):
    """{name: device tensor} -> {name: numpy array}: every tensor copied asynchronously into its own page-locked host block, ONE synchronisation, the
    arrays are views of those blocks (each array owns its block; torch's caching host allocator hands the block out again once the array is
    dropped, so only the first call pays for pinning).  33.8 MB of results of a batch-16 forward: 2.7 ms through `.cpu()` (a pageable destination,
    12.7 GB/s) against 0.9 ms this way."""
    host = {}
    for k, v in tensors.items():
        v = v.detach()
        h = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
        h.copy_(v, non_blocking=True)
        host[k] = h
    torch.cuda.synchronize()
    return {k: h.numpy() for k, h in host.items()}


class _DeferredCheck:
    """Ticket of a sub-graph whose persistent LSTM launches have not been checked yet (InferEngine.speaker_embedding(defer=True))."""

    def __init__(self, eng, fn, a, k, pending, event, host):
        self.eng, self.fn, self.a, self.k, self.pending, self.event, self.host = eng, fn, a, k, pending, event, host

    def ok(self):
        """After (or as) the caller's host sync: did every launch of the sub-graph run to its end?  The ticket owns its page-locked read-back
        block (a second deferred call before this one is redeemed - batch prefetch - reads back into ANOTHER block); it goes back to the
        engine's pool here."""
        self.event.synchronize()
        eng, host = self.eng, self.host
        if host is None:
            raise RuntimeError("a deferred check is redeemed once")
        good = all(int(host[slot][1]) == 0 and int(host[slot][2]) == n_wg for slot, n_wg in self.pending)
        self.host = None
        eng._deferred_host_pool.append(host)
        if eng.persist_lstm_selftest > 0:
            eng.persist_lstm_selftest -= 1
            return False
        return good

    def redo(self):
        """The sub-graph again, launch by launch (same stream); returns its output."""
        eng = self.eng
        eng.persist_lstm_fallbacks += 1
        eng._lstm_retry = True
        try:
            with lib.deterministic_gemm():
                return self.fn(*self.a, **self.k)
        finally:
            eng._lstm_retry = False
            eng._lstm_pending, eng._lstm_copied = [], 0


class InferEngine:
    def __init__(self, dims: Dims = None, device="cuda", seed=1234, params: ParamStore = None, values=None, chunk=50):
        lib.load()
        self.d = dims or Dims()
        self.device = torch.device(device)
        self.seed = seed
        self.params = params if params is not None else ParamStore(self.d, self.device, seed=seed, values=values)
        self.chunk = chunk
        self._keep = []          # keeps temporaries alive until the stream is synchronised
        # persistent launches (csrc/persist_infer.hip, persist_lstm.hip): MSTTS_PERSIST_INFER=0 keeps the launch-per-step drivers
        self.persist_infer = os.environ.get("MSTTS_PERSIST_INFER", "1") != "0"
        self.persist_infer_launches = 0      # decoder loops that ran as one launch
        self.persist_infer_fallbacks = 0     # ... that gave up (bounded waits / no co-residency) and were re-run launch by launch
        self.persist_infer_selftest = 0      # tests: k > 0 makes the launch abort at step k - 1
        self.persist_infer_stamps = None     # bench: 256 x 24 int64 tensor -> per-stage ticks of the next launch
        self.persist_infer_status = None
        self._persist_strikes = 0
        self._persist_off = 0                # decodes left of the cool-down after PERSIST_STRIKES consecutive fallbacks
        self.persist_disabled_decodes = 0
        self.non_persistent_decodes = 0      # decodes at the reference widths whose shape the persistent loop does not cover
        self._warned_shapes = set()
        # recurrent layers around the decoder (encoder BiLSTM, speaker-encoder stack: H = 256; Taco1 BiRNN: H = 128) as ONE persistent launch per
        # layer (csrc/persist_lstm.hip) instead of a launch per step; their control words are read at the forward pass's existing sync points
        self.persist_lstm = os.environ.get("MSTTS_PERSIST_LSTM", "1") != "0"
        self.persist_lstm_launches = 0
        self.persist_lstm_fallbacks = 0      # forward passes re-run without them because a launch gave up
        self.persist_lstm_selftest = 0       # tests: k > 0 makes the next k control-word checks report a launch that gave up
        self._lstm_retry = False
        self._lstm_pending = []              # (slot, workgroups expected to have left in order)
        self._lstm_ctrl = None
        self._spk_stream = None
        self._lstm_ctrl_host = None
        self._lstm_copied = 0
        self._lcache = {}                    # packed recurrent kernels by cell, keyed on ParamStore.version
        self._dcache = {}                    # ... and those both decoder drivers share
        self._pcache = {}                    # variable-derived operands of the persistent decoder, keyed on ParamStore.version

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape):
        n = int(np.prod(shape))
        t = torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=self.device)[:n].view(shape)
        self._keep.append(t)
        return t

    def P(self, name):
        return self.params.p(name)

    def _dense(self, x, rows, cin, cout, kname, bname, out, act=ACT_NONE, lda=None):
        k, ok = self.P(kname)
        if bname is not None:
            b, ob = self.P(bname)
            gemm(x, k, out, rows, cout, cin, lda or cin, cout, cout, bias=b, act=act, b_off=ok, bias_off=ob)
        else:
            gemm(x, k, out, rows, cout, cin, lda or cin, cout, cout, act=act, b_off=ok)

    def _conv_bn(self, x, rows, T, cin, cout, K, prefix, act, bn_prefix=None, conv_name="conv1d"):
        """conv1d 'same' + activation, then inference-mode batch norm (moving statistics)."""
        k, ok = self.P(prefix + conv_name + "/kernel"); b, ob = self.P(prefix + conv_name + "/bias")
        a = self._f(rows, cout)
        gemm(x, k, a, rows, cout, K * cin, cin, cout, cout, bias=b, act=act, win=(T, cin, (K - 1) // 2), b_off=ok, bias_off=ob)
        bnp = bn_prefix or (prefix + "batch_normalization/")
        g, og = self.P(bnp + "gamma"); be, obe = self.P(bnp + "beta")
        mm, omm = self.P(bnp + "moving_mean"); mv, omv = self.P(bnp + "moving_variance")
        y = self._f(rows, cout)
        call("mstts_bn_infer_fwd", ptr(a), ptr(g, og), ptr(be, obe), ptr(mm, omm), ptr(mv, omv), ptr(y), BN_EPS, rows, cout)
        return y

    def _lstm_seq_desc(self, x, B, T, cin, H, cell_prefix, out, out_sb, out_st, out_off=0, lengths=None, reverse=0, residual=None, zc=None, zh=None, fused=True):
        """Descriptor of one ZoneoutLSTMCell over a sequence: inference mode without masks (0.9*new + 0.1*old), training-mode zoneout
        with keep-masks zc / zh [T, B, H] (ZoneoutLSTMCell.py:259-271).  Where the fused cell step covers the shape (no residual) the
        packed recurrent kernel and packed h blocks are attached, so every step is one launch."""
        k, ok = self.P(cell_prefix + "kernel"); b, ob = self.P(cell_prefix + "bias")
        xw = self._f(B * T, 4 * H)
        gemm(x, k, xw, B * T, 4 * H, cin, cin, 4 * H, 4 * H, bias=b, b_off=ok, bias_off=ob)
        q = lib.LstmSeqFwd()
        q.B, q.T, q.H = B, T, H
        q.xw = ptr(xw); q.wh = ptr(k, ok + cin * 4 * H); q.wh_ld = 4 * H
        if reverse and lengths is None:            # tf.reverse_sequence over the full length
            lengths = torch.full((B,), T, dtype=torch.int32, device=self.device)
            self._keep.append(lengths)
        q.lengths = ptr(lengths); q.reverse = reverse; q.zoneout = self.d.zoneout
        q.residual = ptr(residual)
        q.zc, q.zh = ptr(zc), ptr(zh)
        q.out = ptr(out, out_off); q.out_sb = out_sb; q.out_st = out_st
        ch, hh = self._f(T + 1, B, H), self._f(T + 1, B, H)
        q.c_hist, q.h_hist = ptr(ch), ptr(hh)
        L_ = lib.load()
        q.gates_ws = ptr(self._f(int(L_.mstts_lstm_seq_ws_floats(B, H, 0))))
        if fused and residual is None and L_.mstts_cell_fwd_supported(H, H):
            whp = self._f(H * 4 * H)             # (packed per call: the parameters may have been reloaded in between; 1 small launch)
            call("mstts_pack_cell_fwd", ptr(k, ok + cin * 4 * H), 4 * H, ptr(whp), H, H)
            q.wh_p, q.h_p = ptr(whp), ptr(self._f(2 * int(L_.mstts_cell_act_floats(B, H))))
        return q

    # ---- persistent recurrent layers
    def _lstm_persist_args(self, descs, cell_prefixes, cins, B, T, H):
        """Packed kernels (cached), ring, history scratch and a fresh control-word slot for one persistent LSTM launch over `descs`; None when
        the shape / device is not covered or this forward pass is a retry without them."""
        L_, ndir = lib.load(), len(descs)
        if not self.persist_lstm or self._lstm_retry or not L_.mstts_persist_lstm_fwd_supported_n(B, H, ndir):
            return None
        if self._lstm_ctrl is None:
            self._lstm_ctrl = torch.zeros(64, 16, dtype=torch.int32, device=self.device)
            self._lstm_ctrl_host = torch.zeros(64, 16, dtype=torch.int32).pin_memory()
        if len(self._lstm_pending) >= 64:
            return None
        pks = []
        for prefix, cin in zip(cell_prefixes, cins):
            ent = self._lcache.get(prefix)
            if ent is None or ent[0] != self.params.version:
                k, ok = self.P(prefix + "kernel")
                pk = ent[1] if ent is not None else torch.empty(H * 4 * H, dtype=torch.float32, device=self.device)
                call("mstts_persist_lstm_pack_fwd", ptr(k, ok + cin * 4 * H), 4 * H, H, ptr(pk))
                self._lcache[prefix] = ent = (self.params.version, pk)
            pks.append(ent[1])
        empty = lambda n: torch.empty((int(n) + 3) // 4 * 4, dtype=torch.float32, device=self.device)
        xch, hist = empty(L_.mstts_persist_lstm_ws_bytes_n(B, ndir) // 4), empty(L_.mstts_persist_lstm_hist_floats_n(T, B, ndir))
        self._keep.extend((xch, hist))
        slot = len(self._lstm_pending)
        self._lstm_pending.append((slot, ndir * ((B + 31) // 32) * (H // 8)))
        self.persist_lstm_launches += 1
        return pks, ptr(xch), ptr(self._lstm_ctrl, slot * 16), ptr(hist)

    def _lstm_ctrl_copy(self):
        """Enqueue the read-back of the control words of the persistent LSTM launches so far (call in front of a host sync)."""
        if len(self._lstm_pending) > self._lstm_copied:
            self._lstm_ctrl_host.copy_(self._lstm_ctrl, non_blocking=True)
            self._lstm_copied = len(self._lstm_pending)

    def _lstm_verify(self):
        """After a host sync: did every persistent LSTM launch read back so far run to its end?  Raises _PersistRetry otherwise (the
        forward pass is then re-run with the launch-per-step drivers)."""
        if self.persist_lstm_selftest > 0 and self._lstm_copied:
            self.persist_lstm_selftest -= 1
            raise _PersistRetry("persistent LSTM launch 0: self-test")
        for slot, n_wg in self._lstm_pending[:self._lstm_copied]:
            st = self._lstm_ctrl_host[slot]
            if int(st[1]) != 0 or int(st[2]) != n_wg:
                raise _PersistRetry("persistent LSTM launch %d: control words %r" % (slot, st[:3].tolist()))

    def _lstm_seq(self, x, B, T, cin, H, cell_prefix, out, out_sb, out_st, out_off=0, residual=None, **kw):
        # residual wrapper (output = cell output + input, state untouched): with a dense [B, T, H] output the fused steps run without it
        # and one add over the whole sequence follows - the same fp32 sum, 2 launches per step less
        post_add = (residual is not None and out_off == 0 and out_st == H and out_sb == T * H and
                    bool(lib.load().mstts_cell_fwd_supported(H, H)))
        dst = self._f(B, T, H) if post_add else out
        pa = self._lstm_persist_args([None], [cell_prefix], [cin], B, T, H) if (residual is None or post_add) else None
        q = self._lstm_seq_desc(x, B, T, cin, H, cell_prefix, dst, out_sb, out_st, out_off=out_off, residual=None if post_add else residual,
                                fused=pa is None, **kw)
        if pa is not None:
            call("mstts_lstm_seq_fwd_persistent", C.byref(q), ptr(pa[0][0]), pa[1], pa[2], pa[3])
        else:
            call("mstts_lstm_seq_fwd", C.byref(q))
        if post_add:
            call("mstts_add", ptr(dst), ptr(residual), ptr(out), B * T * H)

    def _bilstm_seq(self, x, B, T, cin, H, cell_fmt, out, out_sb, out_st, lengths=None):
        """Forward and backward direction of a bidirectional layer advancing together: one launch per step where the fused form applies
        (mstts_lstm_seq_fwd_pair falls back to two sequential loops otherwise)."""
        pa = self._lstm_persist_args([None, None], [cell_fmt % "fw", cell_fmt % "bw"], [cin, cin], B, T, H)
        qs = [self._lstm_seq_desc(x, B, T, cin, H, cell_fmt % dr, out, out_sb, out_st, out_off=di * H, lengths=lengths, reverse=di, fused=pa is None)
              for di, dr in enumerate(("fw", "bw"))]
        if pa is not None:
            call("mstts_lstm_seq_fwd_pair_persistent", C.byref(qs[0]), C.byref(qs[1]), ptr(pa[0][0]), ptr(pa[0][1]), pa[1], pa[2], pa[3])
        else:
            call("mstts_lstm_seq_fwd_pair", C.byref(qs[0]), C.byref(qs[1]))

    # ------------------------------------------------------------------ sub-graphs
    @lib.deterministic_gemm()
    def _guarded(self, fn, *a, **k):
        """A sub-graph called on its own (outside forward): the control words of its persistent LSTM launches are checked before its
        result is handed out (one stream sync), and the sub-graph is re-run launch by launch if one of them gave up."""
        self._lstm_pending, self._lstm_copied = [], 0
        try:
            out = fn(*a, **k)
            if self._lstm_pending:
                self._lstm_ctrl_copy()
                torch.cuda.current_stream().synchronize()
                self._lstm_verify()
            return out
        except _PersistRetry as e:
            self.persist_lstm_fallbacks += 1
            warnings.warn("multi_speaker_tts_amd: %s; re-running launch by launch" % (e,), RuntimeWarning, stacklevel=3)
            self._lstm_retry = True
            try:
                return fn(*a, **k)
            finally:
                self._lstm_retry = False
                self._lstm_pending, self._lstm_copied = [], 0

    def speaker_embedding(self, spk_mel, masks=None, defer=False):
        """defer=True (the TRAIN step's frozen speaker stack, MSTTS_SV.py:49-56,211): no host sync here.  Returns (embedding, ticket);
        the caller hands the ticket to TrainEngine.forward (batch["_speaker_ticket"]), which redeems it at its one existing sync point and
        re-runs the sub-graph launch by launch - and its own pass - if a persistent launch of the stack gave up."""
        if not defer:
            return self._guarded(self._speaker_embedding, spk_mel, masks=masks)
        return self._deferred(self._speaker_embedding, spk_mel, masks=masks)

    @lib.deterministic_gemm()
    def _deferred(self, fn, *a, **k):
        self._lstm_pending, self._lstm_copied = [], 0
        out = fn(*a, **k)
        pending = list(self._lstm_pending)
        if not pending:
            return out, None
        # one page-locked read-back block PER TICKET (pooled: page-locking is a system call): the copy below is ordered behind this call's launches
        # and in front of the next call's, which reuse the device slots from 0 - tickets may be outstanding together
        pool = self.__dict__.setdefault("_deferred_host_pool", [])
        host = pool.pop() if pool else torch.zeros(64, 16, dtype=torch.int32).pin_memory()
        host.copy_(self._lstm_ctrl, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return out, _DeferredCheck(self, fn, a, k, pending, ev, host)

    def encoder(self, token, token_length, spk):
        return self._guarded(self._encoder, token, token_length, spk)

    def mel_to_spectrogram(self, mel, B, S):
        return self._guarded(self._mel_to_spectrogram, mel, B, S)

    def _speaker_embedding(self, spk_mel, masks=None):
        """[5B,64,80] float32 device tensor -> [B, spk] (MSTTS_SV.py:49-56).  masks: optional {'s_zc_<i>', 's_zh_<i>'} uint8
        keep-masks [64, 5B, cell] - the reference feeds Is_Training into this (frozen) stack too, so a Tacotron2 TRAIN step sees
        stochastic zoneout in the speaker encoder (MSTTS_SV.py:49-56, ZoneoutLSTMCell.py:259-260)."""
        d = self.d
        NB, T, _ = spk_mel.shape
        B = NB // d.spk_samples
        x = self._f(NB * T, d.spk)
        self._dense(spk_mel, NB * T, d.n_mel, d.spk, SPK + "dense/kernel", SPK + "dense/bias", x)
        for i in range(d.spk_lstm_n):
            y = self._f(NB, T, d.spk_lstm)
            self._lstm_seq(x, NB, T, d.spk, d.spk_lstm, SPK_CELL % (i, i), y, T * d.spk_lstm, d.spk_lstm,
                           residual=x if i < d.spk_lstm_n - 1 else None,
                           zc=masks["s_zc_%d" % i] if masks is not None else None, zh=masks["s_zh_%d" % i] if masks is not None else None)
            x = y
        e = self._f(B, d.spk)
        call("mstts_speaker_finalize", ptr(x), ptr(e), B, d.spk_samples, T, d.spk)
        return e

    def _encoder(self, token, token_length, spk, spk_done=None):
        """-> values [B,T,M] (memory masked past Token_Length), keys [B,T,A].  spk_done: event behind the speaker embedding when another stream forms it."""
        d = self.d
        B, T = token.shape
        M, He = d.mem, d.enc_lstm
        emb, oe = self.P("encoder/embedding_variable")
        x = self._f(B * T, d.emb)
        call("mstts_embedding_fwd", ptr(token), ptr(emb, oe), ptr(x), B * T, d.n_tok, d.emb)
        cin = d.emb
        for i in range(d.enc_conv_n):
            x = self._conv_bn(x, B * T, T, cin, d.enc_conv_ch, d.enc_conv_k, "encoder/conv_%d/" % i, ACT_RELU)
            cin = d.enc_conv_ch
        values = self._f(B, T, M)
        self._bilstm_seq(x, B, T, cin, He, ENC_CELL, values, T * M, M, lengths=token_length)
        if spk_done is not None:
            torch.cuda.current_stream().wait_event(spk_done)
        call("mstts_speaker_tile", ptr(spk), ptr(token_length), ptr(values), B, T, M, 2 * He, d.spk)
        keys = self._f(B, T, d.att)
        wm, owm = self.P("attention/memory_layer/kernel")
        gemm(values, wm, keys, B * T, d.att, M, M, d.att, d.att, b_off=owm)
        return values, keys

    @lib.deterministic_gemm()
    def decode(self, values, keys, token_length, masks=None, seed=None, max_steps=None):
        """Free-running decoder.  Returns step-major linear [S,B,n_mel], stop logits [S,B], align [S,B,T]."""
        d = self.d
        B, T, M = values.shape
        H, A, Pn, NM = d.dec_lstm, d.att, d.prenet, d.n_mel
        Smax = (max_steps if max_steps is not None else d.max_inf) + 1
        mk = MaskSet(d, B, T, Smax, False, self.device)
        if masks is not None:
            mk.load(masks)
        else:
            mk.draw(seed if seed is not None else step_seed(self.seed, 0))
        self._keep.append(mk)
        # variable-derived operands shared by both drivers, kept until the variables change: cell-0 kernel with its two context row blocks
        # folded, the folded location filter
        c = self._dcache
        if c.get("version") != self.params.version:
            k0, o0 = self.P(CELL % 0 + "kernel")
            if "w0f" not in c:
                c.update(w0f=torch.zeros((M + H) * 4 * H, dtype=torch.float32, device=self.device).view(M + H, 4 * H),
                         loc_k=torch.zeros(d.att_k * A, dtype=torch.float32, device=self.device).view(d.att_k, A),
                         loc_b=torch.zeros(A, dtype=torch.float32, device=self.device),
                         loc_kt=torch.zeros(A * 36, dtype=torch.float32, device=self.device).view(A, 36) if d.att_k <= 31 else None)
            call("mstts_fold_rows", ptr(k0, o0 + Pn * 4 * H), ptr(c["w0f"]), 2 * M + H, 4 * H, 0, M)
            cp = {}
            for field, name in (("conv_k", "attention_convolution_dense_layer/conv1d/kernel"), ("conv_b", "attention_convolution_dense_layer/conv1d/bias"),
                                ("dense_k", "attention_convolution_dense_layer/dense/kernel")):
                t, o = self.P(LSA + name)
                cp[field] = ptr(t, o)
            call("mstts_lsa_fold_location", cp["conv_k"], cp["conv_b"], cp["dense_k"], ptr(c["loc_k"]), ptr(c["loc_b"]), d.att_k, d.att_ch, A)
            if c["loc_kt"] is not None:
                call("mstts_lsa_filter_by_unit", ptr(c["loc_k"]), ptr(c["loc_kt"]), d.att_k, A)
            c["version"] = self.params.version
        w0f = c["w0f"]
        k0, o0 = self.P(CELL % 0 + "kernel"); b0, ob0 = self.P(CELL % 0 + "bias")
        q = lib.DecoderInfer()
        q.B, q.H, q.P, q.n_mel, q.Smax = B, H, Pn, NM, Smax
        ls = q.lsa
        ls.B, ls.T, ls.A, ls.M, ls.KS, ls.CH = B, T, A, M, d.att_k, d.att_ch
        ls.keys, ls.values, ls.lengths = ptr(keys), ptr(values), ptr(token_length)
        for field, name in (("conv_k", "attention_convolution_dense_layer/conv1d/kernel"), ("conv_b", "attention_convolution_dense_layer/conv1d/bias"),
                            ("dense_k", "attention_convolution_dense_layer/dense/kernel"), ("score_w", "score_layer/weight_w"), ("score_b", "score_layer/bias_b")):
            t, o = self.P(LSA + name)
            setattr(ls, field, ptr(t, o))
        ls.loc_k, ls.loc_b = ptr(c["loc_k"]), ptr(c["loc_b"])
        if c["loc_kt"] is not None:
            ls.loc_kt = ptr(c["loc_kt"])
        for field, name in (("pw0", "decoder/decoder/prenet_0/dense/kernel"), ("pb0", "decoder/decoder/prenet_0/dense/bias"),
                            ("pw1", "decoder/decoder/prenet_1/dense/kernel"), ("pb1", "decoder/decoder/prenet_1/dense/bias"),
                            ("w1", CELL % 1 + "kernel"), ("b1", CELL % 1 + "bias"), ("wq", LSA + "query_layer/kernel"),
                            ("wproj", "decoder/decoder/linear_projection/dense/kernel"), ("bproj", "decoder/decoder/linear_projection/dense/bias")):
            t, o = self.P(name)
            setattr(q, field, ptr(t, o))
        q.pm0, q.pm1, q.prenet_keep = ptr(mk["prenet_drop_0"]), ptr(mk["prenet_drop_1"]), 1 - d.prenet_drop
        q.wx0, q.b0, q.w0f = ptr(k0, o0), ptr(b0, ob0), ptr(w0f)
        q.zoneout = d.zoneout
        # (outputs: every row of the steps that ran is written by whichever driver runs; no clearing pass)
        empty = lambda *shape: torch.empty(int(np.prod(shape)), dtype=torch.float32, device=self.device).view(shape)
        linear, stop, align = empty(Smax, B, NM), empty(Smax, B), empty(Smax, B, T)
        self._keep.extend((linear, stop, align))
        q.linear, q.stop, q.align_hist = ptr(linear), ptr(stop), ptr(align)
        # ONE launch for the whole loop where the shape and the device allow it ...
        S = self._decode_persistent(q, values, w0f, mk, B, T, Smax)
        if S is not None:
            return linear[:S], stop[:S], align[:S], S
        # ... else a launch chain per step
        q.in0, q.in1, q.pj = ptr(self._f(2, B, Pn + M + H)), ptr(self._f(2, B, 2 * H)), ptr(self._f(B, H + M))
        if lib.load().mstts_decoder_infer_fast(B, H, Pn, M, A, NM):
            # weight-streaming path: cell-0 kernel with the prenet rows stacked on the folded [ctx ; h] rows, padded projection
            w0s = self._f(Pn + M + H, 4 * H)
            call("mstts_copy2d", ptr(k0, o0), 4 * H, ptr(w0s), 4 * H, Pn, 4 * H, 0)
            call("mstts_copy2d", ptr(w0f), 4 * H, ptr(w0s, Pn * 4 * H), 4 * H, M + H, 4 * H, 0)
            npad = (NM + 1 + 3) // 4 * 4
            wp_pad = self._f(H + M, npad)
            wpj, owpj = self.P("decoder/decoder/linear_projection/dense/kernel")
            call("mstts_copy2d", ptr(wpj, owpj), NM + 1, ptr(wp_pad), npad, H + M, NM + 1, 0)
            q.w0s, q.wp_pad = ptr(w0s), ptr(wp_pad)
            L_ = lib.load()
            if L_.mstts_cell_fwd_supported(H, Pn + M + H) and L_.mstts_cell_fwd_supported(H, 2 * H):
                # fused cell steps: both cell kernels in the lanes' order + packed activation blocks (csrc/cell.hip)
                w0sp, w1p = self._f((Pn + M + H) * 4 * H), self._f(2 * H * 4 * H)
                k1, o1 = self.P(CELL % 1 + "kernel")
                call("mstts_pack_cell_fwd", ptr(w0s), 4 * H, ptr(w0sp), Pn + M + H, H)
                call("mstts_pack_cell_fwd", ptr(k1, o1), 4 * H, ptr(w1p), 2 * H, H)
                act_p = self._f(2 * int(L_.mstts_cell_act_floats(B, Pn + M + H) + L_.mstts_cell_act_floats(B, 2 * H)))
                q.w0sp, q.w1p, q.act_p = ptr(w0sp), ptr(w1p), ptr(act_p)
                if L_.mstts_lsa_step_qp_supported(T, M, H, npad):
                    # output projection inside the attention launch: the projected values (loop invariant, like the keys) and the
                    # m1 rows of the kernel packed by owner slice
                    wp_own, vp = self._f(int(L_.mstts_lsa_proj_pack_floats())), self._f(B * T, npad)
                    call("mstts_lsa_proj_pack", ptr(wp_pad), npad, H, npad, ptr(wp_own))
                    gemm(values, wp_pad, vp, B * T, npad, M, M, npad, npad, b_off=H * npad)
                    q.wp_own, q.vp = ptr(wp_own), ptr(vp)
        q.c0, q.c1, q.cum = ptr(self._f(2, B, H)), ptr(self._f(2, B, H)), ptr(self._f(2, B, T))
        q.pre_ws = ptr(self._f(int(lib.load().mstts_decoder_infer_ws_floats(B, H, Pn, T, A, NM))))
        # Decoder_Dynamic_Decode stops after the first step at which every row has raised its stop flag
        # (stop_logit >= 0 OR time >= Max_Inference_Length, OR-accumulated; Modules.py:216-219,395,409).
        self._lstm_ctrl_copy()
        finished = np.zeros(B, bool)
        done, S = 0, None
        limit = Smax - 1                         # time index at which the forced stop fires
        while S is None and done < Smax:
            n = min(self.chunk, Smax - done)
            call("mstts_decoder_infer_steps", C.byref(q), done, n)
            st = stop[done:done + n].cpu().numpy()          # synchronises the stream for this chunk
            for i in range(n):
                t = done + i
                finished |= (st[i] >= 0.0) | (t >= limit)
                if finished.all():
                    S = t + 1
                    break
            done += n
        return linear[:S], stop[:S], align[:S], S

    def _decode_persistent(self, q, values, w0f, mk, B, T, Smax):
        """The whole free-running loop as ONE launch (mstts_decoder_infer_persistent).  Returns the number of steps, or None when the
        shape / device is not covered, the persistent plans are cooling down, or the launch gave up - the caller then runs the
        launch-per-step loop, which rewrites every output."""
        d, L_ = self.d, lib.load()
        H, A, Pn, NM, M = d.dec_lstm, d.att, d.prenet, d.n_mel, d.mem
        if not self.persist_infer:
            return None
        if not L_.mstts_persist_infer_supported(B, H, Pn, M, A, T, d.att_k, NM):
            if L_.mstts_persist_infer_supported(1, H, Pn, M, A, 1, d.att_k, NM):      # widths and device fit, this batch shape does not
                self.non_persistent_decodes += 1
                if (B, T) not in self._warned_shapes:
                    self._warned_shapes.add((B, T))
                    warnings.warn("multi_speaker_tts_amd: batch %d x %d tokens is outside the persistent free-running decoder's range "
                                  "(mstts_persist_infer_supported); decoding launch by launch" % (B, T), RuntimeWarning, stacklevel=4)
            return None
        if self._persist_off > 0:
            self._persist_off -= 1
            self.persist_disabled_decodes += 1
            return None
        NP = 84
        pw0, opw0 = self.P("decoder/decoder/prenet_0/dense/kernel"); pb0, opb0 = self.P("decoder/decoder/prenet_0/dense/bias")
        pw1, opw1 = self.P("decoder/decoder/prenet_1/dense/kernel"); pb1, opb1 = self.P("decoder/decoder/prenet_1/dense/bias")
        # what depends on the variables only (padded projection, first prenet layer folded onto its m1 half, kernels in the lanes' order):
        # kept until the variables change (ParamStore.version)
        c = self._pcache
        if c.get("version") != self.params.version:
            alloc = lambda *shape: torch.zeros(int(np.prod(shape)), dtype=torch.float32, device=self.device).view(shape)
            k0, o0 = self.P(CELL % 0 + "kernel"); k1, o1 = self.P(CELL % 1 + "kernel"); wq, oq = self.P(LSA + "query_layer/kernel")
            wpj, owpj = self.P("decoder/decoder/linear_projection/dense/kernel"); bpj, obpj = self.P("decoder/decoder/linear_projection/dense/bias")
            if "wp_pad" not in c:
                c.update(wp_pad=alloc(H + M, NP), bp_pad=alloc(NP), wfm=alloc(H, Pn), bf=alloc(4, Pn), bp4=alloc(4, NP),
                         pk=[torch.empty(int(L_.mstts_persist_pack_floats(i)), dtype=torch.float32, device=self.device) for i in range(3)],
                         wqppk=torch.empty(int(L_.mstts_persist_infer_pack_floats()), dtype=torch.float32, device=self.device),
                         xch=torch.empty(int(L_.mstts_persist_infer_ws_bytes()) // 4, dtype=torch.float32, device=self.device),
                         ctrl=torch.zeros(272, dtype=torch.int32, device=self.device), ctrl_host=torch.zeros(272, dtype=torch.int32).pin_memory())
            call("mstts_copy2d", ptr(wpj, owpj), NM + 1, ptr(c["wp_pad"]), NP, H + M, NM + 1, 0)
            call("mstts_copy2d", ptr(bpj, obpj), NM + 1, ptr(c["bp_pad"]), NP, 1, NM + 1, 0)
            gemm(c["wp_pad"], pw0, c["wfm"], H, Pn, NM, NP, Pn, Pn, b_off=opw0)
            c["bp4"][0].copy_(c["bp_pad"])                # (a 4-row product: row 0 is bp, the other rows are zero)
            gemm(c["bp4"], pw0, c["bf"], 4, Pn, NM, NP, Pn, Pn, bias=pb0, b_off=opw0, bias_off=opb0)
            # (w0f: the caller's folded cell-0 kernel - its values depend on the variables only)
            call("mstts_persist_pack", ptr(w0f), ptr(k1, o1), ptr(wq, oq), ptr(k0, o0), ptr(c["pk"][0]), ptr(c["pk"][1]), ptr(c["pk"][2]))
            call("mstts_persist_infer_pack", ptr(wq, oq), ptr(c["wp_pad"]), NP, ptr(c["wfm"]), ptr(c["wqppk"]))
            c["version"] = self.params.version
        wp_pad, bp_pad, wfm, bf, pk, wqppk, xch, ctrl = (c[k] for k in ("wp_pad", "bp_pad", "wfm", "bf", "pk", "wqppk", "xch", "ctrl"))
        # loop invariants of this batch (see include/mstts.h): projected values, and the first prenet layer applied to them
        vp, u = self._f(B * T, NP), self._f(B * T, Pn)
        gemm(values, wp_pad, vp, B * T, NP, M, M, NP, NP, b_off=H * NP)
        gemm(vp, pw0, u, B * T, Pn, NM, NP, Pn, Pn, b_off=opw0)
        # prenet of the all-zero start frame with the masks of step 0 (Modules.py:178-185)
        zf, pa, pb = self._f(B, NM), self._f(B, Pn), self._f(B, Pn)
        gemm(zf, pw0, pa, B, Pn, NM, NM, Pn, Pn, bias=pb0, act=ACT_RELU, b_off=opw0, bias_off=opb0)
        call("mstts_dropout", ptr(pa), ptr(mk["prenet_drop_0"]), 1 - d.prenet_drop, ptr(pb), B * Pn)
        gemm(pb, pw1, pa, B, Pn, Pn, Pn, Pn, Pn, bias=pb1, act=ACT_RELU, b_off=opw1, bias_off=opb1)
        pre0 = self._f(B, Pn)
        call("mstts_dropout", ptr(pa), ptr(mk["prenet_drop_1"]), 1 - d.prenet_drop, ptr(pre0), B * Pn)
        pd = lib.PersistInferDesc()
        pd.w0pk, pd.w1pk, pd.wqppk = ptr(pk[0]), ptr(pk[1]), ptr(wqppk)
        pd.pre0, pd.bf, pd.u, pd.vp, pd.bp_pad = ptr(pre0), ptr(bf), ptr(u), ptr(vp), ptr(bp_pad)
        pd.xch, pd.ctrl = ptr(xch), ptr(ctrl)
        pd.stamps = ptr(self.persist_infer_stamps) if self.persist_infer_stamps is not None else None
        pd.selftest_fail_step = int(self.persist_infer_selftest)
        pd.near_xcd = int(os.environ.get("MSTTS_PERSIST_NEAR", "1") != "0")
        call("mstts_decoder_infer_persistent", C.byref(q), C.byref(pd))
        self._lstm_ctrl_copy()
        c["ctrl_host"].copy_(ctrl, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        st = c["ctrl_host"].numpy()
        self.persist_infer_status = (int(st[0]), int(st[1]), int(st[2]), int(st[4]), int(st[5]))
        if int(st[1]) != 0 or int(st[2]) != 256:
            self.persist_infer_fallbacks += 1
            self._persist_strikes += 1
            if self._persist_strikes >= 2:
                self._persist_strikes, self._persist_off = 0, 200
                warnings.warn("multi_speaker_tts_amd: two consecutive persistent decoder launches gave up (status %r); decoding launch by "
                              "launch for the next 200 batches" % (self.persist_infer_status,), RuntimeWarning, stacklevel=4)
            return None
        self._persist_strikes = 0
        self.persist_infer_launches += 1
        return int(st[5]) if int(st[5]) > 0 else Smax

    def postnet(self, linear_bsc, B, S):
        d = self.d
        x, cin = linear_bsc, d.n_mel
        for i in range(d.post_n):
            cout = d.post_ch if i < d.post_n - 1 else d.n_mel
            x = self._conv_bn(x, B * S, S, cin, cout, d.post_k, "decoder/conv_%d/" % i, ACT_TANH)
            cin = cout
        mel = self._f(B, S, d.n_mel)
        call("mstts_add", ptr(linear_bsc), ptr(x), ptr(mel), B * S * d.n_mel)
        return mel

    def _mel_to_spectrogram(self, mel, B, S):
        """ConvBank -> Highway -> BiRNN -> Projection (MSTTS_SV.py:100-115), inference mode."""
        d = self.d
        rows, C1 = B * S, d.bank_k * d.bank_ch
        cat = self._f(rows, C1)
        for k in range(1, d.bank_k + 1):
            sfx = bank_suffix(k)
            y = self._conv_bn(mel, rows, S, d.n_mel, d.bank_ch, k, VOC + "convbank_0/", ACT_RELU,
                              bn_prefix=VOC + "convbank_0/batch_normalization%s/" % sfx, conv_name="conv1d%s" % sfx)
            call("mstts_copy2d", ptr(y), d.bank_ch, ptr(cat, (k - 1) * d.bank_ch), C1, rows, d.bank_ch, 0)
        pool = self._f(rows, C1)
        call("mstts_maxpool2_same", ptr(cat), ptr(pool), B, S, C1)
        p1 = self._conv_bn(pool, rows, S, C1, d.proj1_ch, d.proj1_k, VOC + "convbank_0/", ACT_RELU,
                           bn_prefix=VOC + "convbank_0/batch_normalization_8/", conv_name="conv1d_8")
        p2 = self._conv_bn(p1, rows, S, d.proj1_ch, d.n_mel, d.proj2_k, VOC + "convbank_0/", ACT_NONE,
                           bn_prefix=VOC + "convbank_0/batch_normalization_9/", conv_name="conv1d_9")
        x = self._f(rows, d.n_mel)
        call("mstts_add", ptr(mel), ptr(p2), ptr(x), rows * d.n_mel)
        for i in range(d.highway_n):
            hp_, tp_ = self._f(rows, d.n_mel), self._f(rows, d.n_mel)
            self._dense(x, rows, d.n_mel, d.n_mel, VOC + "highway_%d/dense/kernel" % i, VOC + "highway_%d/dense/bias" % i, hp_)
            self._dense(x, rows, d.n_mel, d.n_mel, VOC + "highway_%d/dense_1/kernel" % i, VOC + "highway_%d/dense_1/bias" % i, tp_)
            y = self._f(rows, d.n_mel)
            call("mstts_highway_combine", ptr(hp_), ptr(tp_), ptr(x), ptr(y), rows * d.n_mel)
            x = y
        rnn = self._f(B, S, 2 * d.birnn)
        self._bilstm_seq(x, B, S, d.n_mel, d.birnn, VOC_CELL, rnn, S * 2 * d.birnn, 2 * d.birnn)
        spec = self._f(rows, d.n_spec)
        self._dense(rnn, rows, 2 * d.birnn, d.n_spec, VOC + "dense/kernel", VOC + "dense/bias", spec)
        return spec.view(B, S, d.n_spec)

    # ------------------------------------------------------------------ whole forward
    @lib.deterministic_gemm()
    def forward(self, pattern, masks=None, seed=None, max_steps=None, with_vocoder=True):
        """See _forward.  A persistent LSTM launch that gave up (bounded waits, no co-residency) invalidates what was computed from its output:
        the pass is run again with the launch-per-step drivers."""
        try:
            return self._forward(pattern, masks, seed, max_steps, with_vocoder)
        except _PersistRetry as e:
            self.persist_lstm_fallbacks += 1
            warnings.warn("multi_speaker_tts_amd: %s; re-running the forward pass launch by launch" % (e,), RuntimeWarning, stacklevel=2)
            self._lstm_retry = True
            try:
                return self._forward(pattern, masks, seed, max_steps, with_vocoder)
            finally:
                self._lstm_retry = False

    def _forward(self, pattern, masks=None, seed=None, max_steps=None, with_vocoder=True):
        """pattern: dict with Token [B,T] int32, Token_Length [B] int32 and either Speaker_Embedding_Mel
        [5B,64,80] or Speaker_Embedding [B,spk] (numpy or tensors).  Returns the reference's
        inference_Tensor_Dict as numpy arrays: Linear, Mel, Stop (sigmoid), Attention_History [B,T,S], Spectrogram."""
        self._keep = []
        self._lstm_pending, self._lstm_copied = [], 0
        dev = self.device
        prof = getattr(self, "profile_phases", False)          # tools/infer_bench.py: wall time per phase (adds a synchronisation behind each)
        marks = [("start", time.perf_counter())]

        def mark(name):
            if prof:
                torch.cuda.synchronize()
                marks.append((name, time.perf_counter()))
        t = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))).to(dev, dt).contiguous()
        token, tlen = t(pattern["Token"], torch.int32), t(pattern["Token_Length"], torch.int32)
        B, T = token.shape
        spk_done = None
        if "Speaker_Embedding" in pattern:
            spk = t(pattern["Speaker_Embedding"], torch.float32)
        else:
            spk_mel = t(pattern["Speaker_Embedding_Mel"], torch.float32)
            if SPK_OVERLAP and dev.type == "cuda" and not prof:
                # the speaker stack (three 64-step recurrent layers on 80 rows: 96 workgroups each) and the text encoder (convolutions, BiLSTM: 64 workgroups)
                # meet only at the memory's speaker columns: the stack runs on its own stream beside the encoder.  (Its temporaries live until the end of
                # the pass, self._keep; the control words of its launches are copied behind the join.)
                if self._spk_stream is None:
                    self._spk_stream = torch.cuda.Stream(device=dev)
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(self._spk_stream):
                    self._spk_stream.wait_event(ready)
                    spk = self._speaker_embedding(spk_mel)
                    spk_done = torch.cuda.Event()
                    spk_done.record()
            else:
                spk = self._speaker_embedding(spk_mel)
        mark("uploads + speaker encoder")
        values, keys = self._encoder(token, tlen, spk, spk_done)
        mark("text encoder")
        lin_s, stop_s, align_s, S = self.decode(values, keys, tlen, masks=masks, seed=seed, max_steps=max_steps)
        mark("free-running decoder")
        self._lstm_verify()                  # (decode synchronised the stream behind the read-back of their control words)
        d = self.d
        linear = self._f(B, S, d.n_mel)
        call("mstts_transpose01", ptr(lin_s), ptr(linear), S, B, d.n_mel)
        mel = self.postnet(linear, B, S)
        out = {"Linear": linear, "Mel": mel, "Stop_Logit": stop_s.t().contiguous(), "Attention_History": align_s.permute(1, 2, 0).contiguous(),
               "Speaker_Embedding": spk}
        if with_vocoder:
            out["Spectrogram"] = self._mel_to_spectrogram(mel, B, S)
        self._lstm_ctrl_copy()
        torch.cuda.synchronize()
        self._lstm_verify()
        mark("postnet + Taco1 vocoder")
        res = to_host(out)
        res["Stop"] = 1.0 / (1.0 + np.exp(-res["Stop_Logit"]))
        mark("results to the host")
        if prof:
            self.phase_ms = {b[0]: (b[1] - a[1]) * 1e3 for a, b in zip(marks[:-1], marks[1:])}
        self._keep = []
        return res
