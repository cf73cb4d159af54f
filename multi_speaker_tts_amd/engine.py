"""Train-step engine: one Tacotron2.Train iteration (MSTTS_SV.py:268-273) as an explicit
forward / backward / TF-Adam schedule of libmstts_hip.so calls on one HIP stream.

There is no autograd and no torch compute here: torch tensors are only device buffers.  The
schedule follows the reference graph (Tensor_Generate, MSTTS_SV.py:45-192):
  encoder (embedding -> 3x conv/relu/BN/dropout -> BiLSTM) -> memory/keys -> hoisted prenet ->
  801-step attention decoder loop (native driver) -> projection -> postnet -> losses,
then the exact reverse with hand-written gradients; weight gradients of the recurrent parts are
hoisted out of the time loops into large MFMA GEMMs.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings

import numpy as np
import torch

from . import lib
from .lib import ACT_NONE, ACT_RELU, ACT_TANH, gemm, ptr, call
from .masks import MaskSet, step_seed
from .params import CELL, ENC_CELL, LSA, VOC, Dims, ParamStore, bank_suffix

BN_MOM, BN_EPS = 0.99, 1e-3

MAX_PLANS = 64     # cached workspace sets (one per batch shape): views into ONE arena, a set owns no device memory



def learning_rate(step):
    """MSTTS_SV.py:163-169."""
    from . import Hyper_Parameters as hp
    lr_hp = hp.Train.Learning_Rate
    lr = lr_hp.Initial * lr_hp.Decay_Rate ** ((step - lr_hp.Decay_Start_Step) / lr_hp.Decay_Step)
    return min(max(lr, lr_hp.Min), lr_hp.Initial)


def _split_k(M, N, K, n_cu=256, max_split=16):
    """Reduction split of a weight-gradient GEMM (atomic accumulation into the gradient slab).  128x128 output tiles are
    spread round-robin over the CUs, so a CU runs c = ceil(tiles * sk / n_cu) workgroups of K / sk each: pick the sk with the
    least work on the busiest CU (448 tiles: sk = 4 -> 7 per CU exactly, 127 TFLOP/s, where sk = 2 leaves 3.5 -> 4 per CU,
    107 TFLOP/s; tools/gemm_split_probe.py), with a small per-split charge for the atomics.  Long reductions (the 25 632-row
    products) also weigh HOW MANY workgroups share a CU: one per CU is one wave per SIMD, nothing hides its load / store phases
    (2 560 x 512 x 25 632: sk = 3 -> 240 tiles, 91 TFLOP/s; sk = 9 -> 720 tiles = 3 per CU, 117 TFLOP/s)."""
    tiles = math.ceil(M / 128) * math.ceil(N / 128)
    if K >= 16384:
        ktiles, best, best_cost = K / 32.0, 1, None
        if tiles * max_split < n_cu:             # a handful of tiles (the Taco1 convolution bank: 1..8): split until the chip is covered once
            max_split = min(64, max(max_split, n_cu // tiles))
        for sk in range(1, max_split + 1):
            c = math.ceil(tiles * sk / n_cu)
            cost = c * (ktiles / sk + 6.0) / (0.75 if c == 1 else 0.92 if c == 2 else 1.0) * (1.0 + 0.004 * sk)
            if best_cost is None or cost < best_cost * (1.0 - 1e-9):
                best, best_cost = sk, cost
        return best
    best, best_cost = 1, None
    for sk in range(1, max_split + 1):
        if sk > 1 and K // sk < 512:
            break
        cost = math.ceil(tiles * sk / n_cu) / sk + 0.004 * sk
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = sk, cost
    return best


def _split_k_big(M, N, K, requested, n_cu=256):
    """Reduction split of a weight-gradient product on the 256 x 256-tile kernels (gemm_split_big_kernel / gemm_bf16_big_kernel: one workgroup
    per CU, taken from 160 workgroups on): rounds x (K-tiles per piece + a fixed cost per piece), pieces of at least 1 024 rows.  Returns
    `requested` (the split chosen for 128 x 128 tiles) when no split reaches those kernels."""
    tiles = math.ceil(M / 256) * math.ceil(N / 256)
    if M < 192 or N < 192:
        return requested
    best, best_cost = None, None
    for sk in range(1, 65):
        if sk > 1 and K // sk < 1024:
            break
        if tiles * sk < 160:
            continue
        cost = math.ceil(tiles * sk / n_cu) * (K / sk + 300.0) * (1.0 + 0.01 * sk)      # (+1 % per piece: its atomics onto the shared output)
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = sk, cost
    return best if best is not None else requested


PERSIST_STRIKES = 2          # consecutive steps with a fallback before the persistent plans are switched off ...
PERSIST_COOLDOWN = 200       # ... for this many steps
VOC_OVERLAP = os.environ.get("MSTTS_VOC_OVERLAP", "1") != "0"    # the vocoder conv-bank's statistics side effect (quirk Q20) on its own stream, under the loss and the postnet's backward pass
POSTNET_WGRAD_OVERLAP = os.environ.get("MSTTS_POSTNET_WGRAD_OVERLAP", "1") != "0"   # ... and the postnet's weight-gradient products, beside its data-gradient chain
ENC_TAIL_OVERLAP = os.environ.get("MSTTS_ENC_TAIL_OVERLAP", "1") != "0"   # ... and what follows the encoder's BPTT launch on that stream too, beside the decoder's weight-gradient products
ENC_OVERLAP = os.environ.get("MSTTS_ENC_OVERLAP", "1") != "0"    # the encoder's persistent launches on their own stream, under decoder-side products that do not depend on them


class _WS:
    """Attribute bag of device buffers for one (B, T_enc, L) shape: views into the engine's workspace arena."""


class _LateScalars:
    """Handle of TrainEngine.scalars_async."""

    def __init__(self, event, slot, wr_rate):
        self.event, self.slot, self.wr_rate, self._res = event, slot, wr_rate, None

    def get(self):
        if self._res is None:
            self.event.synchronize()
            s = self.slot.numpy().copy()
            wr = float(s[3]) * self.wr_rate
            self._res = {"Linear_Loss": float(s[0]), "Postnet_Loss": float(s[1]), "Stop_Loss": float(s[2]),
                         "Weight_Regularization_Loss": wr, "Loss": float(s[0] + s[1] + s[2]) + wr}
        return self._res


class _Arena:
    """ONE device buffer that every workspace set is carved from.  The reference feeds a different (T_enc, T_dec) almost every step
    (Feeder.py:111-124,136-175: length-sorted batches, each padded to its own maximum), so a workspace set per shape - ~150 zero-filled
    allocations and ~5 GB at the reference widths - would be built every step.  Only one set is in use at a time; all of them are views of
    the same bytes, laid out by one deterministic walk over the shape (TrainEngine._build_plan), so planning a new shape allocates nothing
    and touches no device memory.  Sized once for the largest shape the engine is told to expect (arena_hint) or grown on demand."""

    ALIGN = 256

    def __init__(self, device):
        self.device, self.buf, self.cap, self.generation = device, None, 0, 0

    def ensure(self, nbytes):
        """True when the arena had to be (re)allocated - every view handed out before is then a view of the OLD buffer."""
        if nbytes <= self.cap:
            return False
        self.buf = None                      # (release first: two 6 GB arenas need not coexist)
        self.cap = int(nbytes)
        self.buf = torch.zeros(self.cap, dtype=torch.uint8, device=self.device)
        self.generation += 1
        return True


class _Carver:
    """One walk over a shape's buffers: count=True only adds up the bytes, else hands out views of the arena."""

    def __init__(self, arena, count):
        self.arena, self.count, self.off = arena, count, 0
        self.zero = []                       # views that must read as zeros when their set is activated (flag / counter workspaces)
        self.overflow = False                # carving ran past the arena's end: the walk goes on counting, its views are None from there on

    def take(self, shape, dtype=torch.float32, zero=False):
        n = int(np.prod(shape)) if not isinstance(shape, int) else int(shape)
        size = torch.empty(0, dtype=dtype).element_size()
        nbytes = (n * size + _Arena.ALIGN - 1) // _Arena.ALIGN * _Arena.ALIGN
        off, self.off = self.off, self.off + max(nbytes, _Arena.ALIGN)
        if self.count or self.off > self.arena.cap:
            self.overflow = self.overflow or not self.count
            return None
        t = self.arena.buf[off:off + n * size].view(dtype)
        t = t.view(shape) if not isinstance(shape, int) else t
        if zero:
            self.zero.append(t)
        return t


class TrainEngine:
    def __init__(self, dims: Dims = None, device="cuda", seed=1234, rank=0, world=1, values=None,
                 update_vocoder_bn=True, use_l1=None, wr_rate=None, adam=None, recurrent_dtype=None, gemm_dtype=None, fuse_query=True,
                 arena_hint=None, deterministic=None):
        """recurrent_dtype: 'f32' (default; BASELINE config 2) or 'bf16' (config 3: the decoder's recurrent products run on
        bf16 copies of the fp32 master weights with fp32 accumulation).
        arena_hint: (B, T_enc, L) of the largest batch to expect - the workspace arena is sized for it once, so no later shape allocates.
        deterministic (default: MSTTS_DETERMINISTIC=1 in the environment): train_step runs with every summation order fixed - no reduction
        cut the engine or the library would choose (split_k = 1 everywhere, mstts_gemm_deterministic), column sums / the embedding scatter /
        the attention parameter gradients in their one-add-per-element forms - so two runs from the same state end bit-identical
        (tests/test_gpu_model.py::test_deterministic_training_is_bit_reproducible) where the persistent launches run (the reference widths); the
        launch-per-step fallback loops keep K-cuts with atomics of their own.  A debugging mode: 121 ms per step at the headline shape (2.9 x)."""
        lib.load()
        self.d = dims or Dims()
        self.device = torch.device(device)
        self.seed, self.rank, self.world = seed, rank, world
        self.update_vocoder_bn = update_vocoder_bn
        self.fuse_query = bool(fuse_query)     # query projection inside the attention launch when the geometry allows (mstts_lsa_step_fwd_q)
        from . import Hyper_Parameters as hp           # None = the drop-in hyper parameters (MSTTS_SV.py:138-176)
        self.use_l1 = bool(hp.Train.Use_L1_Loss) if use_l1 is None else use_l1
        self.wr_rate = float(hp.Train.Weight_Regularization_Rate) if wr_rate is None else wr_rate
        self.adam = (hp.Train.ADAM.Beta1, hp.Train.ADAM.Beta2, hp.Train.ADAM.Epsilon) if adam is None else adam
        self.params = ParamStore(self.d, self.device, seed=seed, values=values)
        self._plans = {}          # workspace sets (views of the arena) keyed by batch shape, least recently used first (at most MAX_PLANS kept)
        self._arena = _Arena(self.device)
        self._active_plan = None  # the set whose activation fills were run last (a set of another shape has used the same bytes since otherwise)
        self._pinned = {}         # page-locked read-back blocks of the persistent launches' control words, shared by every set
        self.arena_hint = arena_hint
        self.deterministic = (os.environ.get("MSTTS_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        self.arena_poison = os.environ.get("MSTTS_ARENA_POISON", "0") == "1"     # tests: NaN over a set's whole extent whenever it is activated
        self.global_step = 0
        d = self.d
        # packed / derived weights refreshed after every optimizer step
        H, M = d.dec_lstm, d.mem
        self.w0f = self._f(M + H, 4 * H)
        self.proj_ld = (d.n_mel + 1 + 3) // 4 * 4
        self.wp_pad = self._f(H + M, self.proj_ld)
        self.bp_pad = self._f(self.proj_ld)
        self.dwp_pad = self._f(H + M, self.proj_ld)
        self.dw0f = self._f(M + H, 4 * H)
        self.loc_k, self.loc_b, self.d_loc_k = self._f(d.att_k, d.att), self._f(d.att), self._f(d.att_k, d.att)
        self.loc_kt = self._f(d.att, 36) if d.att_k <= 31 else None       # the folded filter by unit (forward attention step)
        # fused cell steps (csrc/cell.hip): the two decoder cell kernels in the lanes' consumption order
        lb = lib.load()
        self.fused_cells = bool(lb.mstts_cell_fwd_supported(H, M + H)) and bool(lb.mstts_cell_fwd_supported(H, 2 * H))
        self.w0p = self._f((M + H) * 4 * H) if self.fused_cells else None
        self.w1p = self._f(2 * H * 4 * H) if self.fused_cells else None
        self.w0p16 = self.w1p16 = None          # bf16 copies for the fused cell steps of the config-3 mode (allocated below)
        # packed kernels of the BPTT data-gradient products (csrc/skinny.hip, PACKED form)
        self.bwd_splits = (int(lb.mstts_skinny_bwd_splits(M + H, 4 * H)), int(lb.mstts_skinny_bwd_splits(2 * H, 4 * H)), int(lb.mstts_skinny_bwd_splits(H, d.att)))
        ok = lambda R, sp: sp > 0 and R % 32 == 0
        self.w0f_bp = self._f((M + H) * 4 * H) if ok(M + H, self.bwd_splits[0]) else None
        self.w1_bp = self._f(2 * H * 4 * H) if ok(2 * H, self.bwd_splits[1]) else None
        self.wq_bp = self._f(H * d.att) if ok(H, self.bwd_splits[2]) else None
        He = d.enc_lstm
        self.enc_whp = {dr: self._f(He * 4 * He) for dr in ("fw", "bw")} if lb.mstts_cell_fwd_supported(He, He) else None
        self.wq_t = self._f(d.att * H) if d.att == 128 else None         # query kernel as [A/4, H, 4] (fused query-layer data gradient)
        # persistent decoder loop (csrc/persist.hip): reference widths; MSTTS_PERSIST=0 keeps the launch-per-step loop.  fp32 recurrent products
        # (BASELINE config 2), or - round 5 - bf16 ones when the whole step runs config-3 arithmetic (recurrent_dtype AND gemm_dtype "bf16":
        # the bf16 instantiation forms the prenet rows' product itself, as a bf16 product like the hoisted one it replaces; with fp32
        # contractions around bf16 loops that product would have to stay fp32, and that combination keeps the launch-per-step loops)
        rdt, gdt = (recurrent_dtype or "f32").lower(), (gemm_dtype or "f32").lower()
        self.persist_bf16 = rdt == "bf16" and gdt == "bf16" and d.prenet == 256 and os.environ.get("MSTTS_PERSIST_BF16", "1") != "0"
        self.persist = (os.environ.get("MSTTS_PERSIST", "1") != "0" and (rdt == "f32" or self.persist_bf16)
                        and bool(lb.mstts_persist_fwd_supported(1, H, M, d.att, 1, d.att_k)))
        self.exact_f32_products = False      # True: no split products inside the persistent forward either (see forward(); include/mstts.h, mstts_persist_desc.pre)
        self.persist_fallbacks = 0           # sequences that had to be re-run on the launch-per-step path
        # Adaptive policy: a persistent launch that gives up costs its rendezvous bound plus the slow loop, and something that holds CUs
        # (another process, a profiler, a collective that outlives its slot) will do so again next step.  After PERSIST_STRIKES
        # consecutive steps with a fallback the persistent plans are switched off for PERSIST_COOLDOWN steps (warned once), then probed
        # again.  persist_disabled_steps counts the steps run that way; non_persistent_plans the shapes planned without them although
        # the widths are the reference's (the T_enc / batch cliff of the persistent kernels).
        self._moving_snapshot = None
        self._persist_strikes = 0
        self._persist_off = 0                # steps left of the cool-down
        self._persist_warned = False
        self.persist_disabled_steps = 0
        self.non_persistent_plans = 0
        self._warned_shapes = set()
        self.persist_selftest = 0            # tests: k > 0 makes the persistent forward launch abort at step k - 1
        self.persist_bwd_selftest = 0        # ... and the persistent BPTT launch at its k-th step
        self.persist_stamps = None           # bench: 256 x 16 int64 tensor -> per-stage ticks of the next persistent launch
        self.persist_bwd = self.persist and os.environ.get("MSTTS_PERSIST_BWD", "1") != "0" and bool(lb.mstts_persist_bwd_supported(1, H, M, d.att, 1, d.att_k))
        self.persist_bwd_fallbacks = 0
        self.speaker_ticket_redos = 0        # forward passes re-run because a persistent launch of the speaker stack in front of them gave up
        self.trace_events, self.bptt_end_event = False, None
        self.collective_redos = 0            # data parallel: backward passes re-run because ANOTHER rank's persistent launch gave up
        self.persist_bwd_stamps = None
        if self.persist:
            self.pk = [self._f(int(lb.mstts_persist_pack_floats(i))) for i in range(3)]
        # persistent encoder BiLSTM (csrc/persist_lstm.hip): all T steps of both directions in one launch each way; MSTTS_PERSIST_ENC=0
        # keeps the launch-per-step pair drivers
        self.persist_enc = os.environ.get("MSTTS_PERSIST_ENC", "1") != "0" and bool(lb.mstts_persist_lstm_supported(1, He))
        self.persist_enc_fallbacks = 0
        if self.persist_enc:
            n = int(lb.mstts_persist_lstm_pack_floats())
            self.enc_pk = {dr: (self._f(n), self._f(n)) for dr in ("fw", "bw")}          # (forward order, BPTT order)
        if self.device.type == "cuda":
            self._side = torch.cuda.Stream(device=self.device)                # status words -> pinned host memory, the job-wide verdict's exchange
            self._enc_stream = torch.cuda.Stream(device=self.device)          # the encoder's persistent launches (forward / loss_and_backward)
            # train_step: the vocoder conv-bank's BN-statistics side effect (read by nothing in the step) runs on the encoder's stream too, which is idle between
            # the encoder's forward and BPTT launches.  NOT a stream of its own: HIP maps streams onto four hardware queues, and in a process with an RCCL
            # group a third engine stream landed on the main stream's queue - every launch serialized, nothing gained (profiles/r06_ab_side_chains.txt)
            self._voc_stream = self._enc_stream
        if self.persist_bwd:
            self.pkb = [self._f(int(lb.mstts_persist_bwd_pack_floats(i))) for i in range(3)]
        self.flip = {}
        self.big_tiles = os.environ.get("MSTTS_SPLIT_K_BIG", "1") != "0"      # weight-gradient splits chosen for the 256 x 256-tile kernels
        self._derived_stale = True
        self.gemm_dtype = (gemm_dtype or "f32").lower()
        if self.gemm_dtype not in ("f32", "bf16"):
            raise ValueError("gemm_dtype must be 'f32' or 'bf16'")
        self.recurrent_dtype = (recurrent_dtype or "f32").lower()
        if self.recurrent_dtype not in ("f32", "bf16"):
            raise ValueError("recurrent_dtype must be 'f32' or 'bf16'")
        self.bf = None
        if self.recurrent_dtype == "bf16" and lb.mstts_cell_fwd_bf16_supported(H, M + H) and lb.mstts_cell_fwd_bf16_supported(H, 2 * H):
            self.w0p16 = torch.zeros((M + H) * 4 * H, dtype=torch.int16, device=self.device)
            self.w1p16 = torch.zeros(2 * H * 4 * H, dtype=torch.int16, device=self.device)
        if self.recurrent_dtype == "bf16":
            sp = (C.c_int32 * 6)()
            if not lib.load().mstts_decoder_bf16_splits(H, M, d.att, sp):
                raise ValueError("bf16 recurrent products need dec_lstm, mem and att widths that are multiples of 64")
            i16 = lambda n: torch.zeros(n, dtype=torch.int16, device=self.device)
            self.bf = {"splits": list(sp), "w0f_f": i16((M + H) * 4 * H), "w1_f": i16(2 * H * 4 * H), "wq_f": i16(H * d.att),
                       "w0f_b": i16((M + H) * 4 * H), "w1_b": i16(2 * H * 4 * H), "wq_b": i16(H * d.att)}

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape):
        n = int(np.prod(shape))
        return torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=self.device)[:n].view(shape)

    def _gemm(self, *a, exact=False, **k):
        """Every dense / conv contraction of the step.  gemm_dtype 'bf16' (BASELINE config 3) rounds both operands to bf16 for the
        matrix cores (fp32 accumulate, fp32 master weights / activations / gradients in memory); exact=True keeps the few
        contractions that have no dense-layer counterpart in the reference graph (d_values from the alignments, the vocoder's
        statistics side effect) in fp32 in either mode."""
        bf = self.gemm_dtype == "bf16" and not exact
        if self.deterministic and k.get("split_k", 1) > 1:
            # one piece: the pieces of a cut product meet in atomic adds, whose order is not fixed.  The callers of cut products add onto a
            # cleared (or, with accumulate, a live) output, so the single piece accumulates too.
            k["split_k"], k["accumulate"] = 1, True
        if k.get("split_k", 1) > 1 and self.big_tiles:       # (the split was chosen for 128 x 128 tiles; the large contractions run 256 x 256 ones)
            big = _split_k_big(a[3], a[4], a[5], None)
            if big is not None:
                k["split_k"] = max(2, big)
            elif bf:
                # config 3, a product the 256 x 256-tile kernel does not take: mstts_gemm_bf16 honours a caller's cut exactly, and the cut above was
                # chosen for the fp32 kernel's one-workgroup-per-CU tiles - hand the product over uncut and let the library cut it for its own
                # (two workgroups per CU).  It adds its pieces onto the output, which every caller of a cut product has zeroed (the gradient slab,
                # dw0f, dwp_pad) or accumulates into.
                k["split_k"], k["accumulate"] = 1, True
        return gemm(*a, bf16=bf, **k)

    def P(self, name):
        return self.params.p(name)

    def G(self, name):
        return self.params.g(name)

    def _ensure_fallback_packs(self):
        """Packed kernels of the launch-per-step loops (decoder cells forward / data-gradient products backward, encoder cells, the
        transposed query kernel of the fused query gradient), refreshed at most once per optimizer step."""
        if not getattr(self, "_fallback_packs_stale", True):
            return
        self._fallback_packs_stale = False
        d = self.d
        H, M = d.dec_lstm, d.mem
        k1, o1 = self.P(CELL % 1 + "kernel"); wq_, oq_ = self.P(LSA + "query_layer/kernel")
        if self.fused_cells and self.bf is None:
            call("mstts_pack_cell_fwd", ptr(self.w0f), 4 * H, ptr(self.w0p), M + H, H)
            call("mstts_pack_cell_fwd", ptr(k1, o1), 4 * H, ptr(self.w1p), 2 * H, H)
        if self.w0p16 is not None:
            call("mstts_pack_cell_fwd_bf16", ptr(self.w0f), 4 * H, ptr(self.w0p16), M + H, H)
            call("mstts_pack_cell_fwd_bf16", ptr(k1, o1), 4 * H, ptr(self.w1p16), 2 * H, H)
        if self.w0f_bp is not None:
            call("mstts_pack_skinny_bwd", ptr(self.w0f), 4 * H, ptr(self.w0f_bp), M + H, 4 * H, self.bwd_splits[0])
        if self.w1_bp is not None:
            call("mstts_pack_skinny_bwd", ptr(k1, o1), 4 * H, ptr(self.w1_bp), 2 * H, 4 * H, self.bwd_splits[1])
        if self.wq_bp is not None:
            call("mstts_pack_skinny_bwd", ptr(wq_, oq_), d.att, ptr(self.wq_bp), H, d.att, self.bwd_splits[2])
        if self.enc_whp is not None:
            cin_e, He = d.enc_conv_ch, d.enc_lstm
            for dr in ("fw", "bw"):
                ke, oke = self.P(ENC_CELL % dr + "kernel")
                call("mstts_pack_cell_fwd", ptr(ke, oke + cin_e * 4 * He), 4 * He, ptr(self.enc_whp[dr]), He, He)
        if self.wq_t is not None:
            call("mstts_transpose01", ptr(wq_, oq_), ptr(self.wq_t), H, d.att // 4, 4)      # [H, A/4, 4] -> [A/4, H, 4]
        if self.bf is not None:              # bf16 copies of the master weights for the launch-per-step bf16 loops, in the lanes' consumption order
            A_ = d.att
            f0, f1, fq, b0, b1, bq = self.bf["splits"]
            call("mstts_pack_bf16_fwd", ptr(self.w0f), 4 * H, ptr(self.bf["w0f_f"]), M + H, 4 * H, f0)
            call("mstts_pack_bf16_fwd", ptr(k1, o1), 4 * H, ptr(self.bf["w1_f"]), 2 * H, 4 * H, f1)
            call("mstts_pack_bf16_fwd", ptr(wq_, oq_), A_, ptr(self.bf["wq_f"]), H, A_, fq)
            call("mstts_pack_bf16_bwd", ptr(self.w0f), 4 * H, ptr(self.bf["w0f_b"]), M + H, 4 * H, b0)
            call("mstts_pack_bf16_bwd", ptr(k1, o1), 4 * H, ptr(self.bf["w1_b"]), 2 * H, 4 * H, b1)
            call("mstts_pack_bf16_bwd", ptr(wq_, oq_), A_, ptr(self.bf["wq_b"]), H, A_, bq)

    def refresh_derived(self):
        """Folded cell-0 kernel (context rows appear twice) and the 4-column padded
        projection kernel.  Must run after the variables change."""
        d, ps = self.d, self.params
        H, M, Pn = d.dec_lstm, d.mem, d.prenet
        k0, o0 = self.P(CELL % 0 + "kernel")
        call("mstts_fold_rows", ptr(k0, o0 + Pn * 4 * H), ptr(self.w0f), 2 * M + H, 4 * H, 0, M)
        # the launch-per-step kernels' packed copies (cells, data-gradient products, encoder cells, transposed query kernel): every step where
        # those loops run, on demand where the persistent launches do the work (_ensure_fallback_packs: ~0.1 ms of packing per step otherwise)
        self._fallback_packs_stale = True
        if not (self.persist and self.persist_bwd and self.persist_enc):
            self._ensure_fallback_packs()
        k1, o1 = self.P(CELL % 1 + "kernel"); wq_, oq_ = self.P(LSA + "query_layer/kernel")
        if self.persist_enc:
            cin_e, He = d.enc_conv_ch, d.enc_lstm
            for dr in ("fw", "bw"):
                ke, oke = self.P(ENC_CELL % dr + "kernel")
                call("mstts_persist_lstm_pack", ptr(ke, oke + cin_e * 4 * He), 4 * He, ptr(self.enc_pk[dr][0]), ptr(self.enc_pk[dr][1]))
        if self.persist:
            call("mstts_persist_pack", ptr(self.w0f), ptr(k1, o1), ptr(wq_, oq_), ptr(k0, o0), ptr(self.pk[0]), ptr(self.pk[1]), ptr(self.pk[2]))
        if self.persist_bwd:
            call("mstts_persist_bwd_pack", ptr(self.w0f), ptr(k1, o1), ptr(wq_, oq_), ptr(self.pkb[0]), ptr(self.pkb[1]), ptr(self.pkb[2]))
        wp, owp = self.P("decoder/decoder/linear_projection/dense/kernel")
        bp, obp = self.P("decoder/decoder/linear_projection/dense/bias")
        nm1 = d.n_mel + 1
        self.wp_pad.zero_()
        self.bp_pad.zero_()
        call("mstts_copy2d", ptr(wp, owp), nm1, ptr(self.wp_pad), self.proj_ld, H + M, nm1, 0)
        call("mstts_copy2d", ptr(bp, obp), nm1, ptr(self.bp_pad), self.proj_ld, 1, nm1, 0)
        ck, ock = self.P(LSA + "attention_convolution_dense_layer/conv1d/kernel"); cb, ocb = self.P(LSA + "attention_convolution_dense_layer/conv1d/bias")
        dk, odk = self.P(LSA + "attention_convolution_dense_layer/dense/kernel")
        call("mstts_lsa_fold_location", ptr(ck, ock), ptr(cb, ocb), ptr(dk, odk), ptr(self.loc_k), ptr(self.loc_b), d.att_k, d.att_ch, d.att)
        if self.loc_kt is not None:
            call("mstts_lsa_filter_by_unit", ptr(self.loc_k), ptr(self.loc_kt), d.att_k, d.att)
        self._derived_stale = False

    def _pin(self, name, n):
        if name not in self._pinned:
            self._pinned[name] = torch.zeros(n, dtype=torch.int32).pin_memory()
        return self._pinned[name]

    def plan(self, B, Te, L):
        """The workspace set of a batch shape: views into the arena (no allocation once the arena covers the shape; `arena_hint`).  Sets of
        different shapes alias each other - exactly one is in use at a time, the one the last plan() call returned."""
        key = (B, Te, L)
        w = self._plans.get(key)
        if w is not None and w.arena_generation == self._arena.generation:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        else:
            # one walk in the usual case (the arena covers the shape: carve at once); a walk that runs past the arena's end keeps counting, the
            # arena grows to what it counted (to the hint's size the first time) and the shape is walked again
            cv = _Carver(self._arena, self._arena.buf is None)
            w, need = self._build_plan(B, Te, L, cv)
            if cv.count or cv.overflow:
                if self._arena.buf is None and self.arena_hint is not None:
                    hb, ht, hl = self.arena_hint
                    need = max(need, self._build_plan(max(B, hb), max(Te, ht), max(L, hl), _Carver(self._arena, True))[1])
                if self._arena.ensure(need):
                    self._plans.clear()                      # (sets carved from the old buffer stay valid for whoever holds them; they are not handed out again)
                    self._active_plan = None
                w = self._build_plan(B, Te, L, _Carver(self._arena, False))[0]
            while len(self._plans) >= MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = w
        if self._active_plan is not w:
            self._activate(w)
        return w

    def _activate(self, w):
        """A set of another shape may have used these bytes since this set ran last.  Every buffer of a set is written before it is read
        within a step (the library clears what it needs cleared: initial states, rings, control words) - except the few flag / counter
        workspaces collected in w.zero_on_activate, cleared here.  arena_poison (tests: MSTTS_ARENA_POISON=1): NaN over the set's whole
        extent first, so that anything that does rely on stale or zero-initialised memory fails a parity test loudly."""
        if self.arena_poison and w.extent_bytes >= 4:
            call("mstts_fill", ptr(self._arena.buf), float("nan"), w.extent_bytes // 4)
        for t in w.zero_on_activate:
            t.zero_()
        self._active_plan = w

    def _build_plan(self, B, Te, L, cv):
        """One deterministic walk over every buffer of the shape: with a counting carver it returns (None, bytes), else (set, bytes)."""
        d = self.d
        f = lambda *shape, zero=False: cv.take(int(np.prod(shape)), zero=zero) if cv.count else cv.take(shape if len(shape) > 1 else int(shape[0]), zero=zero)
        w = _WS()
        S = L + 1
        w.B, w.Te, w.L, w.S = B, Te, L, S
        H, M, A, Pn, He = d.dec_lstm, d.mem, d.att, d.prenet, d.enc_lstm
        w.masks = MaskSet(d, B, Te, S, True, self.device, rank=self.rank, alloc=lambda n: cv.take(n, dtype=torch.uint8))
        # encoder
        w.emb = f(B * Te, d.emb)
        w.enc_a = [f(B * Te, d.enc_conv_ch) for _ in range(d.enc_conv_n)]
        w.enc_y = [f(B * Te, d.enc_conv_ch) for _ in range(d.enc_conv_n)]
        w.enc_mean = [f(d.enc_conv_ch) for _ in range(d.enc_conv_n)]
        w.enc_rstd = [f(d.enc_conv_ch) for _ in range(d.enc_conv_n)]
        w.enc_xw = {dr: f(B * Te, 4 * He) for dr in ("fw", "bw")}
        w.enc_c = {dr: f(Te + 1, B, He) for dr in ("fw", "bw")}
        w.enc_h = {dr: f(Te + 1, B, He) for dr in ("fw", "bw")}
        w.enc_acts = {dr: f(Te, B, 4 * He) for dr in ("fw", "bw")}
        w.enc_craw = {dr: f(Te, B, He) for dr in ("fw", "bw")}
        w.enc_gates = {dr: f(int(lib.load().mstts_lstm_seq_ws_floats(B, He, 0))) for dr in ("fw", "bw")}
        w.enc_hp = {dr: f(2 * int(lib.load().mstts_cell_act_floats(B, He))) for dr in ("fw", "bw")} if self.enc_whp is not None else None
        w.values = f(B, Te, M)
        w.keys = f(B, Te, A)
        # decoder
        w.frames = f(S, B, d.n_mel)
        w.pre_a = [f(S * B, Pn) for _ in range(d.prenet_n)]      # relu outputs
        w.pre_d = [f(S * B, Pn) for _ in range(d.prenet_n)]      # after dropout
        w.xw0 = f(S, B, 4 * H)
        w.in0, w.in1, w.pj = f(S + 1, B, M + H), f(S + 1, B, 2 * H), f(S, B, H + M)
        w.c0, w.c1 = f(S + 1, B, H), f(S + 1, B, H)
        if w.in1 is not None and w.c0 is not None and w.c1 is not None:
            # slot S of these histories is written by no loop (in1[S] = [m0_S | h1_{S-1}]: there is no step S; the packed operand blocks do not carry
            # the state behind the last step) and read by none; cleared on activation so that whoever looks at a whole history sees numbers
            cv.zero.extend([w.in1[S], w.c0[S], w.c1[S]])
        w.acts0, w.acts1 = f(S, B, 4 * H), f(S, B, 4 * H)
        w.craw0, w.craw1 = f(S, B, H), f(S, B, H)
        w.q_hist, w.align_hist, w.cum_hist = f(S, B, A), f(S, B, Te), f(S + 1, B, Te)
        lb = lib.load()
        ng, nq = C.c_int64(0), C.c_int64(0)
        lb.mstts_decoder_train_ws_floats(B, H, M, A, C.byref(ng), C.byref(nq))
        w.energy_ws_floats = int(lib.load().mstts_lsa_step_q_ws_bytes(B, Te)) // 4 + 2      # room for the in-launch query exchange
        w.gates_ws, w.energy_ws, w.q_ws = f(int(ng.value), zero=True), f(w.energy_ws_floats, zero=True), f(int(nq.value), zero=True)
        w.act_p = f(2 * int(lb.mstts_cell_act_floats(B, M + H) + lb.mstts_cell_act_floats(B, 2 * H))) if self.fused_cells else None
        w.persist_enc = self.persist_enc and bool(lb.mstts_persist_lstm_supported(B, He))
        if w.persist_enc:
            w.enc_xch = f(int(lb.mstts_persist_lstm_ws_bytes()) // 4)
            w.enc_ctrl = cv.take(16, dtype=torch.int32)
            w.enc_ctrl_host = self._pin("enc_ctrl_host", 16)
            w.enc_ctrl_host_b = self._pin("enc_ctrl_host_b", 16)
            w.enc_hist = f(int(lb.mstts_persist_lstm_hist_floats(Te)))        # packed per-step inputs + history of the persistent forward
            w.enc_bws = f(int(lb.mstts_persist_lstm_bwd_floats(Te)))
            w.enc_hist_valid = False
        w.persist = self.persist and bool(lb.mstts_persist_fwd_supported(B, H, M, A, Te, d.att_k))
        if self.persist and not w.persist and not cv.count and not cv.overflow:
            # the device and the widths admit the persistent launches, this batch shape does not: ~1.7x slower loop - say so, once per shape
            self.non_persistent_plans += 1
            if (B, Te) not in self._warned_shapes:
                self._warned_shapes.add((B, Te))
                warnings.warn("multi_speaker_tts_amd: batch %d x %d tokens is outside the persistent decoder kernels' range "
                              "(mstts_persist_fwd_supported); this shape runs the launch-per-step loops" % (B, Te), RuntimeWarning, stacklevel=3)
        if w.persist:
            w.xch = f(int(lb.mstts_persist_fwd_ws_bytes()) // 4)
            w.pctrl = cv.take(272, dtype=torch.int32)
            w.pctrl_host = self._pin("pctrl_host", 272)
            w.pdesc = lib.PersistDesc()
        w.persist_bwd = w.persist and self.persist_bwd and bool(lb.mstts_persist_bwd_supported(B, H, M, A, Te, d.att_k))
        if w.persist_bwd:
            w.xch_b = f(int(lb.mstts_persist_bwd_ws_bytes()) // 4)
            w.pctrl_b = cv.take(272, dtype=torch.int32)
            w.pctrl_b_host = self._pin("pctrl_b_host", 272)
            w.pdesc_b = lib.PersistDesc()
            w.opk = f(int(lb.mstts_persist_opk_floats(S)))         # the cell updates' BPTT operands, packed by owner (instead of acts / craw / c)
        w.opk_valid = False
        w.proj = f(S, B, self.proj_ld)
        w.linear, w.stop = f(B, S, d.n_mel), f(B, S)
        # postnet
        chans = [d.post_ch] * (d.post_n - 1) + [d.n_mel]
        w.post_a = [f(B * S, c) for c in chans]
        w.post_y = [f(B * S, c) for c in chans]
        w.post_mean = [f(c) for c in chans]
        w.post_rstd = [f(c) for c in chans]
        w.mel_out = f(B, S, d.n_mel)
        w.bn_ws = f(2 * max(d.post_ch, d.enc_conv_ch, d.bank_k * d.bank_ch, d.proj1_ch, d.n_mel, d.emb))
        # vocoder conv-bank (BN update side effect)
        if self.update_vocoder_bn:
            w.v_tmp = f(B * S, d.bank_ch)
            w.v_tmp2 = f(B * S, d.bank_ch)
            w.v_cat = f(B * S, d.bank_k * d.bank_ch)
            w.v_pool = f(B * S, d.bank_k * d.bank_ch)
            w.v_p1 = f(B * S, d.proj1_ch)
            w.v_p1y = f(B * S, d.proj1_ch)
            w.v_p2 = f(B * S, d.n_mel)
            w.v_p2y = f(B * S, d.n_mel)
            w.v_stat = f(2 * max(d.bank_ch, d.proj1_ch, d.n_mel))
            w.v_bn_ws = f(2 * max(d.bank_ch, d.proj1_ch, d.n_mel))          # (its own column-sum workspace: the chain may run beside the main stream's BN calls)
        # losses
        w.scalars = f(4)
        w.d_linear, w.d_post, w.d_stop = f(B, S, d.n_mel), f(B, S, d.n_mel), f(B, S)
        # backward
        w.post_dz = [f(B * S, c) for c in chans]
        w.post_dx = f(B * S, max(d.post_ch, d.n_mel))
        w.post_dx2 = f(B * S, max(d.post_ch, d.n_mel))
        w.d_proj = f(S, B, self.proj_ld)
        w.d_pj = f(S, B, H + M)
        w.dg0, w.dg1 = f(S, B, 4 * H), f(S, B, 4 * H)
        w.dq_hist, w.de_hist = f(S, B, A), f(S, B, Te)
        w.d_in0_parts = int(lb.mstts_decoder_train_bwd_parts(H, M))
        w.d_in0 = f(w.d_in0_parts, S, B, M + H)
        w.dec_bwd_ws = f(int(lb.mstts_decoder_train_bwd_ws_floats(B, H, M, A, Te, d.att_ch)), zero=True)
        w.lsa_param_ws = f((int(lb.mstts_lsa_param_bwd_ws_floats(B, Te, S)) + 3) // 4 * 4)   # partial blocks of the attention parameter gradients
        w.d_pre = f(S * B, Pn)
        w.d_pre2 = f(S * B, Pn)
        w.d_keys = f(B, Te, A)
        w.d_values = f(B, Te, M)
        w.enc_dgs = {dr: f(Te, B, 4 * He) for dr in ("fw", "bw")}
        w.enc_dgp = {dr: f(B, Te, 4 * He) for dr in ("fw", "bw")}
        w.enc_bwd_ws = {dr: f(int(lb.mstts_lstm_seq_ws_floats(B, He, 1))) for dr in ("fw", "bw")}
        w.enc_dy = f(B * Te, max(d.enc_conv_ch, d.emb))
        w.enc_dz = f(B * Te, d.enc_conv_ch)
        w.enc_dx = f(B * Te, max(d.enc_conv_ch, d.emb))
        # descriptors with stable addresses
        w.job_flag = cv.take(1, dtype=torch.int32)
        w.job_flag_host = self._pin("job_flag_host", 1)
        w.dec = lib.DecoderTrain()
        w.dec_b = lib.DecoderTrainBwd()
        w.zero_on_activate = cv.zero
        w.extent_bytes = cv.off
        w.arena_generation = self._arena.generation
        return (None if (cv.count or cv.overflow) else w), cv.off

    # ------------------------------------------------------------------ conv blocks
    def _conv_fwd(self, x, x_off, rows, T, cin, cout, K, kname, bname, out, act):
        k, ok = self.P(kname)
        b, ob = self.P(bname)
        self._gemm(x, k, out, rows, cout, K * cin, cin, cout, cout, bias=b, act=act, win=(T, cin, (K - 1) // 2),
             a_off=x_off, b_off=ok, bias_off=ob)

    def _bn_fwd(self, prefix, a, y, mean, rstd, mask, keep, rows, C, ws):
        g, og = self.P(prefix + "gamma"); b, ob = self.P(prefix + "beta")
        mm, omm = self.P(prefix + "moving_mean"); mv, omv = self.P(prefix + "moving_variance")
        call("mstts_bn_train_fwd", ptr(a), ptr(g, og), ptr(b, ob), ptr(mm, omm), ptr(mv, omv), ptr(y), ptr(mean), ptr(rstd),
             ptr(mask), float(keep), BN_MOM, BN_EPS, rows, C, ptr(ws))

    def _conv_block_bwd(self, dy, x_in, a, mean, rstd, mask, keep, act, prefix, rows, T, cin, cout, K, dz, dx, wgrad_stream=None):
        """BN(+dropout)+activation+conv backward.  dy: grad of the block output; returns nothing;
        dx (or None) receives the input gradient; parameter grads accumulate into the grad slab.
        wgrad_stream: run the kernel's weight-gradient product (read by nothing before Adam) on that stream, behind this block's dz."""
        g, og = self.P(prefix + "batch_normalization/gamma")
        gg, ogg = self.G(prefix + "batch_normalization/gamma")
        gb, ogb = self.G(prefix + "batch_normalization/beta")
        gbias, ogbias = self.G(prefix + "conv1d/bias")
        call("mstts_bn_train_bwd", ptr(dy), ptr(a), ptr(g, og), ptr(mean), ptr(rstd), ptr(mask), float(keep), act, ptr(dz),
             ptr(gg, ogg), ptr(gb, ogb), ptr(gbias, ogbias), rows, cout, ptr(self._bnws))
        gk, ogk = self.G(prefix + "conv1d/kernel")
        pad = (K - 1) // 2
        if wgrad_stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(wgrad_stream):
                wgrad_stream.wait_event(ev)
                self._gemm(x_in, dz, gk, K * cin, cout, rows, cin, cout, cout, trans_a=True, win=(T, cin, pad),
                     split_k=max(2, _split_k(K * cin, cout, rows)), c_off=ogk)
        else:
            self._gemm(x_in, dz, gk, K * cin, cout, rows, cin, cout, cout, trans_a=True, win=(T, cin, pad),
                 split_k=max(2, _split_k(K * cin, cout, rows)), c_off=ogk)
        if dx is not None:
            k, ok = self.P(prefix + "conv1d/kernel")
            key = (prefix, K, cin, cout)
            if key not in self.flip:
                self.flip[key] = self._f(K, cout, cin)
            wt = self.flip[key]
            call("mstts_conv_kernel_flip", ptr(k, ok), ptr(wt), K, cin, cout)
            self._gemm(dz, wt, dx, rows, cin, K * cout, cout, cin, cin, win=(T, cout, K - 1 - pad))

    # ------------------------------------------------------------------ forward
    def forward(self, batch, w, seed=None, masks=None, _redo=False, allowed=None, overlap_vocoder=False):
        """Forward pass.  The persistent launches (encoder BiLSTM, decoder loop) are enqueued WITHOUT waiting for their control words; the
        words of both are read once, behind the rest of the pass (one host sync per pass).  If either launch gave up, the BN moving statistics
        are put back to their state at the start of the pass and the whole pass is run again with the launch-per-step loops (_redo).
        allowed: the fallback policy's decision for this optimizer step (train_step takes it ONCE per step with _persist_begin_step and
        hands it down); None = a forward pass driven on its own (tests, tools) advances the policy itself.
        overlap_vocoder: train_step only - the vocoder conv-bank's statistics side effect (quirk Q20: ~50 small launches, 1.0 ms, read by nothing in
        the step) goes to its own stream and loss_and_backward joins it in front of the decoder's BPTT launch (which needs every CU); a forward
        pass driven on its own keeps it on the caller's stream."""
        d, ps = self.d, self.params
        B, Te, L, S = w.B, w.Te, w.L, w.S
        H, M, A, Pn, He = d.dec_lstm, d.mem, d.att, d.prenet, d.enc_lstm
        w.voc_done = None
        if self._active_plan is not w and not _redo:         # (a set planned earlier, while a set of another shape has used the arena since)
            self._activate(w)
        if self._derived_stale:
            self.refresh_derived()
        allowed = False if _redo else (self._persist_begin_step() if allowed is None else bool(allowed))
        w.persist_now = bool(w.persist) and allowed
        w.persist_bwd_now = bool(getattr(w, "persist_bwd", False)) and allowed
        speculative = (allowed and (w.persist_now or bool(getattr(w, "persist_enc", False)))) or (not _redo and "_speaker_ticket" in batch)
        if speculative:
            # the pass runs on the persistent launches' outputs before their status words are known; a launch that gave up leaves junk there,
            # and the BN layers would fold that junk into their MOVING statistics - one update per train step is the reference's behaviour
            # (Modules.py:37-40 update ops), so the whole range (one contiguous slice of the non-trainable slab) is put back before the re-run
            n_mov = self.params.n_moving
            if self._moving_snapshot is None or self._moving_snapshot.numel() != n_mov:
                self._moving_snapshot = torch.empty(n_mov, dtype=torch.float32, device=self.device)
            self._moving_snapshot.copy_(self.params.frozen[:n_mov])
        # (the decoder-side masks are drawn under the encoder's persistent launch when that runs on its own stream)
        late_masks = masks is None and bool(getattr(w, "persist_enc", False)) and allowed and ENC_OVERLAP
        mask_seed = seed if seed is not None else step_seed(self.seed, self.global_step)
        if masks is not None:
            w.masks.load(masks)
        else:
            w.masks.draw(mask_seed, only=(lambda n: n.startswith("enc_")) if late_masks else None)
        mk = w.masks
        self._bnws = w.bn_ws
        tok, tlen = batch["Token"], batch["Token_Length"]
        mel, mlen, spk = batch["Mel"], batch["Mel_Length"], batch["Speaker_Embedding"]
        w.batch = batch
        # ---- encoder (Modules.py:15-73)
        emb, oe = self.P("encoder/embedding_variable")
        call("mstts_embedding_fwd", ptr(tok), ptr(emb, oe), ptr(w.emb), B * Te, d.n_tok, d.emb)
        x, cin = w.emb, d.emb
        for i in range(d.enc_conv_n):
            pre = "encoder/conv_%d/" % i
            self._conv_fwd(x, 0, B * Te, Te, cin, d.enc_conv_ch, d.enc_conv_k, pre + "conv1d/kernel", pre + "conv1d/bias", w.enc_a[i], ACT_RELU)
            self._bn_fwd(pre + "batch_normalization/", w.enc_a[i], w.enc_y[i], w.enc_mean[i], w.enc_rstd[i],
                         mk["enc_conv_drop_%d" % i], 1 - d.conv_drop, B * Te, d.enc_conv_ch, w.bn_ws)
            x, cin = w.enc_y[i], d.enc_conv_ch
        seqs = []
        for di, dr in enumerate(("fw", "bw")):
            k, ok = self.P(ENC_CELL % dr + "kernel"); b, ob = self.P(ENC_CELL % dr + "bias")
            self._gemm(x, k, w.enc_xw[dr], B * Te, 4 * He, cin, cin, 4 * He, 4 * He, bias=b, b_off=ok, bias_off=ob)
            q = lib.LstmSeqFwd()
            q.B, q.T, q.H = B, Te, He
            q.xw = ptr(w.enc_xw[dr]); q.wh = ptr(k, ok + cin * 4 * He); q.wh_ld = 4 * He
            q.lengths = ptr(tlen); q.reverse = di; q.zoneout = d.zoneout
            q.zc = ptr(mk["enc_zc_" + dr]); q.zh = ptr(mk["enc_zh_" + dr])
            q.out = ptr(w.values, di * He); q.out_sb = Te * M; q.out_st = M
            q.c_hist = ptr(w.enc_c[dr]); q.h_hist = ptr(w.enc_h[dr]); q.acts = ptr(w.enc_acts[dr]); q.c_raw = ptr(w.enc_craw[dr])
            q.gates_ws = ptr(w.enc_gates[dr])
            if self.enc_whp is not None:                 # fused steps: packed recurrent kernel + packed h blocks
                q.wh_p, q.h_p = ptr(self.enc_whp[dr]), ptr(w.enc_hp[dr])
            seqs.append(q)
        enc_ticket, enc_done = None, None
        w.enc_hist_valid = bool(getattr(w, "persist_enc", False)) and allowed
        if w.enc_hist_valid and ENC_OVERLAP:
            # (the persistent launch occupies 64 of the 256 CUs for 0.34 ms: on its own stream, under the decoder's hoisted prenet, which does not depend on it)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._enc_stream):
                self._enc_stream.wait_event(ready)
                enc_ticket = self._enc_persistent(w, "mstts_lstm_seq_fwd_pair_persistent", seqs, 0, 64)   # (status read at the end of the pass)
                enc_done = torch.cuda.Event()
                enc_done.record()
        elif w.enc_hist_valid:
            enc_ticket = self._enc_persistent(w, "mstts_lstm_seq_fwd_pair_persistent", seqs, 0, 64)
        if not w.enc_hist_valid:
            self._ensure_fallback_packs()
            call("mstts_lstm_seq_fwd_pair", C.byref(seqs[0]), C.byref(seqs[1]))     # both directions advance together: one launch per step
        if late_masks:
            w.masks.draw(mask_seed, only=lambda n: not n.startswith("enc_"))
        # ---- hoisted prenet over all S frames (Modules.py:239-255) and cell-0 input product
        call("mstts_shift_frames", ptr(mel), ptr(w.frames), B, L, d.n_mel)
        x, cin = w.frames, d.n_mel
        for i in range(d.prenet_n):
            k, ok = self.P("decoder/decoder/prenet_%d/dense/kernel" % i); b, ob = self.P("decoder/decoder/prenet_%d/dense/bias" % i)
            self._gemm(x, k, w.pre_a[i], S * B, Pn, cin, cin, Pn, Pn, bias=b, act=ACT_RELU, b_off=ok, bias_off=ob)
            call("mstts_dropout", ptr(w.pre_a[i]), ptr(mk["prenet_drop_%d" % i]), 1 - d.prenet_drop, ptr(w.pre_d[i]), S * B * Pn)
            x, cin = w.pre_d[i], Pn
        if enc_done is not None:
            torch.cuda.current_stream().wait_event(enc_done)
        # ---- memory = [encoder | speaker], masked past Token_Length; keys = values . W_mem
        if batch.get("_speaker_event") is not None:          # (the product surface forms the embedding on another stream, beside everything above)
            torch.cuda.current_stream().wait_event(batch["_speaker_event"])
        call("mstts_speaker_tile", ptr(spk), ptr(tlen), ptr(w.values), B, Te, M, 2 * He, d.spk)
        wm, owm = self.P("attention/memory_layer/kernel")
        self._gemm(w.values, wm, w.keys, B * Te, A, M, M, A, A, b_off=owm)
        k0, o0 = self.P(CELL % 0 + "kernel"); b0, ob0 = self.P(CELL % 0 + "bias")
        # cell-0 input product xw0 = prenet . W0[:P] + b0: inside the persistent launch (fp32 mode, 256-wide prenet), else hoisted here
        # (exact_f32_products: every product of the loop on the f32-input MFMA - the launch's round-4 form, which keeps the prenet rows' product
        #  hoisted; selected together with mstts_gemm_split3(0) by whoever wants IEEE fp32 products everywhere, e.g. bench.py's strict leg)
        w.fold_prenet = w.persist_now and Pn == 256 and (self.persist_bf16 or (self.gemm_dtype == "f32" and not self.exact_f32_products
                                                                                and os.environ.get("MSTTS_PERSIST_FOLD", "1") != "0"))
        xw0_product = lambda: self._gemm(x, k0, w.xw0, S * B, 4 * H, Pn, Pn, 4 * H, 4 * H, bias=b0, b_off=o0, bias_off=ob0)
        if not w.fold_prenet:
            xw0_product()
        # ---- decoder loop
        dec = w.dec
        dec.B, dec.S, dec.H, dec.P = B, S, H, Pn
        ls = dec.lsa
        ls.B, ls.T, ls.A, ls.M, ls.KS, ls.CH = B, Te, A, M, d.att_k, d.att_ch
        ls.keys, ls.values, ls.lengths = ptr(w.keys), ptr(w.values), ptr(tlen)
        for field, name in (("conv_k", "attention_convolution_dense_layer/conv1d/kernel"), ("conv_b", "attention_convolution_dense_layer/conv1d/bias"),
                            ("dense_k", "attention_convolution_dense_layer/dense/kernel"), ("score_w", "score_layer/weight_w"), ("score_b", "score_layer/bias_b")):
            t, o = self.P(LSA + name)
            setattr(ls, field, ptr(t, o))
        ls.loc_k, ls.loc_b, ls.loc_kt = ptr(self.loc_k), ptr(self.loc_b), ptr(self.loc_kt)
        k1, o1 = self.P(CELL % 1 + "kernel"); b1, ob1 = self.P(CELL % 1 + "bias"); wq, oq = self.P(LSA + "query_layer/kernel")
        dec.xw0, dec.w0f, dec.w1, dec.b1, dec.wq = ptr(w.xw0), ptr(self.w0f), ptr(k1, o1), ptr(b1, ob1), ptr(wq, oq)
        dec.zc0, dec.zh0, dec.zc1, dec.zh1 = ptr(mk["dec_zc_0"]), ptr(mk["dec_zh_0"]), ptr(mk["dec_zc_1"]), ptr(mk["dec_zh_1"])
        dec.zoneout = d.zoneout
        if self.bf is not None:
            dec.bf_w0f_f, dec.bf_w1_f, dec.bf_wq_f = ptr(self.bf["w0f_f"]), ptr(self.bf["w1_f"]), ptr(self.bf["wq_f"])
            dec.bf_w0f_b, dec.bf_w1_b, dec.bf_wq_b = ptr(self.bf["w0f_b"]), ptr(self.bf["w1_b"]), ptr(self.bf["wq_b"])
        dec.w0f_bp, dec.w1_bp, dec.wq_bp, dec.wq_t = ptr(self.w0f_bp), ptr(self.w1_bp), ptr(self.wq_bp), ptr(self.wq_t)
        if self.fused_cells and self.bf is None:
            dec.w0p, dec.w1p, dec.act_p = ptr(self.w0p), ptr(self.w1p), ptr(w.act_p)
        if self.w0p16 is not None and self.bf is not None and w.act_p is not None:
            dec.w0p16, dec.w1p16, dec.act_p = ptr(self.w0p16), ptr(self.w1p16), ptr(w.act_p)
        dec.chains = 1
        dec.energy_ws_floats = w.energy_ws_floats if self.fuse_query else 0
        for nm in ("in0", "in1", "pj", "c0", "c1", "acts0", "acts1", "craw0", "craw1", "q_hist", "align_hist", "cum_hist", "gates_ws", "energy_ws", "q_ws"):
            setattr(dec, nm, ptr(getattr(w, nm)))
        ev = None
        if w.persist_now:
            # ONE launch for all S steps; the launch-per-step loop below is the fallback when the 256 workgroups were not co-resident
            # or a bounded wait expired (ctrl words, checked after the rest of the forward pass is enqueued - no bubble on the device)
            pd = w.pdesc
            pd.w0pk, pd.w1pk, pd.wqpk, pd.xch, pd.ctrl = ptr(self.pk[0]), ptr(self.pk[1]), ptr(self.pk[2]), ptr(w.xch), ptr(w.pctrl)
            pd.stamps = ptr(self.persist_stamps) if self.persist_stamps is not None else None
            pd.opk = ptr(w.opk) if w.persist_bwd_now else None
            w.opk_valid = pd.opk is not None
            pd.selftest_fail_step = int(self.persist_selftest)
            pd.near_xcd = int(os.environ.get("MSTTS_PERSIST_NEAR", "1") != "0")
            pd.recurrent_bf16 = int(self.persist_bf16)
            pd.pre, pd.b0 = (ptr(x), ptr(b0, ob0)) if w.fold_prenet else (None, None)
            call("mstts_decoder_train_fwd_persistent", C.byref(dec), C.byref(pd))
            # the launch's control words -> page-locked memory, ON THIS STREAM right behind the launch, then an event the host waits for at the
            # end of the pass.  (Until round 6 a side stream did the copy behind a wait for the launch.  A stream's wait is a barrier packet in its
            # hardware queue, and HIP maps streams onto a few hardware queues: in a process with more streams - an RCCL group - the side stream
            # shared the main stream's queue, and every kernel enqueued behind such a barrier waited with it: profiles/r06_one_rank_rccl_timeline_before.txt,
            # the decoder's weight-gradient products sat out the encoder's BPTT launch, +1.0 ms per step.)
            w.pctrl_host.copy_(w.pctrl, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            w.opk_valid = False
            self._ensure_fallback_packs()
            call("mstts_decoder_train_fwd", C.byref(dec))
        self._forward_tail(w, overlap_vocoder=overlap_vocoder and VOC_OVERLAP and self.device.type == "cuda")
        dec_ok = enc_ok = True
        if ev is not None:
            ev.synchronize()
            st = w.pctrl_host
            dec_ok = int(st[1]) == 0 and int(st[2]) == 256
            if not dec_ok:
                self.persist_fallbacks += 1
                self.persist_last_status = (int(st[0]), int(st[1]), int(st[2]))
        if enc_ticket is not None:
            enc_ok = self._enc_check(w, enc_ticket)
        spk_ticket = None if _redo else batch.pop("_speaker_ticket", None)
        if spk_ticket is not None and not spk_ticket.ok():
            # the frozen speaker stack in front of this pass (MSTTS_SV.py:49-56) ran as persistent launches whose words are read only now:
            # one of them gave up, the embedding this pass consumed is junk - recompute it launch by launch, then run the pass again
            batch["Speaker_Embedding"].copy_(spk_ticket.redo())
            self.speaker_ticket_redos += 1
            enc_ok = False
        if not (dec_ok and enc_ok):
            self._step_fell_back = True
            torch.cuda.current_stream().synchronize()
            if w.voc_done is not None:                   # (the side chain writes moving statistics too)
                w.voc_done.synchronize()
            self.params.frozen[:self.params.n_moving].copy_(self._moving_snapshot)
            return self.forward(batch, w, seed=seed, masks=masks, _redo=True, overlap_vocoder=overlap_vocoder)
        return w

    def _enc_persistent(self, w, entry, seqs, which, n_wg):
        """One persistent launch for all steps of both encoder directions (forward: which = 0, BPTT: 1).  Nothing waits here: the launch's
        control words are copied to pinned host memory behind it on the same stream; the returned ticket is redeemed with _enc_check at the
        end of the pass."""
        extra = (ptr(w.enc_hist),) if which == 0 else (ptr(w.enc_hist), ptr(w.enc_bws))
        call(entry, C.byref(seqs[0]), C.byref(seqs[1]), ptr(self.enc_pk["fw"][which]), ptr(self.enc_pk["bw"][which]), ptr(w.enc_xch), ptr(w.enc_ctrl), *extra)
        if getattr(self, "persist_enc_selftest", 0):              # tests: this encoder launch "gave up" - on the DEVICE, where the job-wide verdict reads it too
            self.persist_enc_selftest -= 1
            w.enc_ctrl[1:2].fill_(3)
        host = w.enc_ctrl_host if which == 0 else w.enc_ctrl_host_b
        host.copy_(w.enc_ctrl, non_blocking=True)            # on the launching stream, behind the launch (see forward(): no side-stream barrier)
        done = torch.cuda.Event()
        done.record()
        return done, host, n_wg

    def _enc_check(self, w, ticket):
        """True when the encoder launch of the ticket ran to its end (False: the caller re-runs the pass with the launch-per-step pair)."""
        done, host, n_wg = ticket
        done.synchronize()
        if int(host[1]) != 0 or int(host[2]) != n_wg:
            self.persist_enc_fallbacks += 1
            self._step_fell_back = True
            return False
        return True

    def _persist_begin_step(self):
        """Adaptive fallback policy, called once per OPTIMIZER STEP (train_step; a forward pass driven on its own calls it itself): closes the books on the previous step (a step in which any persistent
        launch gave up is a strike; PERSIST_STRIKES in a row start a cool-down) and says whether this step may use the persistent
        launches."""
        if getattr(self, "_step_fell_back", False):
            self._persist_strikes += 1
            if self._persist_strikes >= PERSIST_STRIKES:
                self._persist_off = PERSIST_COOLDOWN
                self._persist_strikes = 0
                if not self._persist_warned:
                    self._persist_warned = True
                    warnings.warn("multi_speaker_tts_amd: %d consecutive steps fell back from the persistent launches (status %r: something "
                                  "else holds compute units); running the launch-per-step loops for %d steps before probing again"
                                  % (PERSIST_STRIKES, getattr(self, "persist_last_status", None), PERSIST_COOLDOWN), RuntimeWarning, stacklevel=4)
        elif getattr(self, "_step_was_persistent", False):
            self._persist_strikes = 0
        self._step_fell_back = False
        if self._persist_off > 0:
            self._persist_off -= 1
            self.persist_disabled_steps += 1
            self._step_was_persistent = False
            return False
        self._step_was_persistent = True
        return True

    def unpack_history(self, w):
        """Packed cell-update operands of the persistent forward -> the row-major histories acts0/1, craw0/1, c0/1 (what the
        launch-per-step BPTT and the tests read)."""
        if getattr(w, "opk_valid", False):
            call("mstts_persist_unpack_history", ptr(w.opk), C.byref(w.dec))

    def _forward_tail(self, w, overlap_vocoder=False):
        """Everything behind the decoder loop: projection, postnet, residual, the vocoder's statistics side effect."""
        d = self.d
        B, S = w.B, w.S
        H, M = d.dec_lstm, d.mem
        mk = w.masks
        # ---- projection (Modules.py:309-321) on all steps at once, then batch-major linear/stop
        self._gemm(w.pj, self.wp_pad, w.proj, S * B, self.proj_ld, H + M, H + M, self.proj_ld, self.proj_ld, bias=self.bp_pad)
        call("mstts_unpack_proj", ptr(w.proj), self.proj_ld, ptr(w.linear), ptr(w.stop), B, S, d.n_mel)
        # ---- postnet (Modules.py:121-143) + residual
        x, cin = w.linear, d.n_mel
        for i in range(d.post_n):
            pre = "decoder/conv_%d/" % i
            cout = d.post_ch if i < d.post_n - 1 else d.n_mel
            self._conv_fwd(x, 0, B * S, S, cin, cout, d.post_k, pre + "conv1d/kernel", pre + "conv1d/bias", w.post_a[i], ACT_TANH)
            self._bn_fwd(pre + "batch_normalization/", w.post_a[i], w.post_y[i], w.post_mean[i], w.post_rstd[i],
                         mk["post_drop_%d" % i], 1 - d.conv_drop, B * S, cout, w.bn_ws)
            x, cin = w.post_y[i], cout
        call("mstts_add", ptr(w.linear), ptr(x), ptr(w.mel_out), B * S * d.n_mel)
        if self.update_vocoder_bn and overlap_vocoder:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._voc_stream):
                self._voc_stream.wait_event(ready)
                self._vocoder_bn_update(w)
                w.voc_done = torch.cuda.Event()
                w.voc_done.record()
        elif self.update_vocoder_bn:
            self._vocoder_bn_update(w)

    def _join_vocoder(self, w):
        """The caller's stream waits for the side chains of this pass that run on the encoder's stream (if any): the vocoder's statistics side effect of
        the forward pass, the postnet's weight-gradient products."""
        if getattr(w, "voc_done", None) is not None:
            torch.cuda.current_stream().wait_event(w.voc_done)
            w.voc_done = None
        if getattr(w, "side_wgrads", False):
            done = torch.cuda.Event()
            done.record(self._enc_stream)
            torch.cuda.current_stream().wait_event(done)
            w.side_wgrads = False

    def _vocoder_bn_update(self, w):
        """Quirk Q20: the train op also runs the vocoder conv-bank's BN update ops on the predicted mel."""
        d = self.d
        B, S = w.B, w.S
        rows = B * S
        for k in range(1, d.bank_k + 1):
            sfx = bank_suffix(k)
            kk, ok = self.P(VOC + "convbank_0/conv1d%s/kernel" % sfx); b, ob = self.P(VOC + "convbank_0/conv1d%s/bias" % sfx)
            self._gemm(w.mel_out, kk, w.v_tmp, rows, d.bank_ch, k * d.n_mel, d.n_mel, d.bank_ch, d.bank_ch, bias=b, act=ACT_RELU,
                 win=(S, d.n_mel, (k - 1) // 2), b_off=ok, bias_off=ob, exact=True)
            self._bn_fwd(VOC + "convbank_0/batch_normalization%s/" % sfx, w.v_tmp, w.v_tmp2, w.v_stat, w.v_stat[d.bank_ch:], None, 1.0, rows, d.bank_ch, w.v_bn_ws)
            call("mstts_copy2d", ptr(w.v_tmp2), d.bank_ch, ptr(w.v_cat, (k - 1) * d.bank_ch), d.bank_k * d.bank_ch, rows, d.bank_ch, 0)
        C1 = d.bank_k * d.bank_ch
        call("mstts_maxpool2_same", ptr(w.v_cat), ptr(w.v_pool), B, S, C1)
        kk, ok = self.P(VOC + "convbank_0/conv1d_8/kernel"); b, ob = self.P(VOC + "convbank_0/conv1d_8/bias")
        self._gemm(w.v_pool, kk, w.v_p1, rows, d.proj1_ch, d.proj1_k * C1, C1, d.proj1_ch, d.proj1_ch, bias=b, act=ACT_RELU,
             win=(S, C1, (d.proj1_k - 1) // 2), b_off=ok, bias_off=ob, exact=True)
        self._bn_fwd(VOC + "convbank_0/batch_normalization_8/", w.v_p1, w.v_p1y, w.v_stat, w.v_stat[d.proj1_ch:], None, 1.0, rows, d.proj1_ch, w.v_bn_ws)
        kk, ok = self.P(VOC + "convbank_0/conv1d_9/kernel"); b, ob = self.P(VOC + "convbank_0/conv1d_9/bias")
        self._gemm(w.v_p1y, kk, w.v_p2, rows, d.n_mel, d.proj2_k * d.proj1_ch, d.proj1_ch, d.n_mel, d.n_mel, bias=b,
             win=(S, d.proj1_ch, (d.proj2_k - 1) // 2), b_off=ok, bias_off=ob, exact=True)
        self._bn_fwd(VOC + "convbank_0/batch_normalization_9/", w.v_p2, w.v_p2y, w.v_stat, w.v_stat[d.n_mel:], None, 1.0, rows, d.n_mel, w.v_bn_ws)

    # ------------------------------------------------------------------ loss + backward
    def loss_and_backward(self, w, grad_scale=1.0, on_ready=None, on_abort=None, agree=None, agree_async=None, _redo=False):
        """Loss and backward pass.  Like forward(): the persistent launches (decoder BPTT, encoder BPTT) are enqueued without waiting for their
        control words, which are read once at the end of the pass; if either gave up the whole pass is run again with the launch-per-step
        loops (it starts by clearing the gradient slab).  on_abort: called before that re-run (train_step: wait for the collectives that the
        abandoned pass has already started on the slab).
        agree: data-parallel runs - callable(passed) -> bool, the MINIMUM of `passed` over the ranks (GradAllReduce.agree).  The collectives
        on_ready started have already mixed this pass's gradients into every rank's slab, so keeping or re-running the pass must be ONE
        decision of the whole job: either every rank keeps its pass, or every rank drains (on_abort) and runs the pass again - each rank
        then issues the same sequence of collectives, and a rank whose own launches were healthy throws away the junk a peer contributed
        (`collective_redos` counts the passes re-run because a PEER's launch gave up).
        agree_async: the same decision without a host round trip (GradAllReduce.agree_async, RCCL only): callable(flag) that turns the int32
        device word `flag` into its minimum over the ranks, in place, on the current stream.  The word is formed on the side stream from the
        launches' control words (mstts_persist_status) right behind them, exchanged on its own communicator and read - with the control
        words - at the pass's one host sync, which therefore still falls while the hoisted products run.  Takes precedence over `agree`."""
        d, ps = self.d, self.params
        B, Te, L, S = w.B, w.Te, w.L, w.S
        H, M, A, Pn, He = d.dec_lstm, d.mem, d.att, d.prenet, d.enc_lstm
        mk, batch = w.masks, w.batch
        tok, tlen, mel, mlen = batch["Token"], batch["Token_Length"], batch["Mel"], batch["Mel_Length"]
        ps.grad.zero_()
        w.scalars.zero_()
        call("mstts_tts_loss_fwd_bwd", ptr(w.linear), ptr(w.mel_out), ptr(mel), ptr(w.stop), ptr(mlen), B, S, d.n_mel,
             int(self.use_l1), float(grad_scale), ptr(w.scalars), ptr(w.d_linear), ptr(w.d_post), ptr(w.d_stop))
        call("mstts_l2_loss_acc", ptr(ps.train), ptr(ps.wd_mask), ps.n_train, ptr(w.scalars, 3))
        # ---- postnet backward.  In a train step (a vocoder side chain is running) the five weight-gradient products go to the encoder's stream as well, behind
        # that chain: the ≈1 ms of small launches between this stream's data-gradient products (BN backward, column sums, kernel flips) then run beside a
        # product instead of alone on the chip.  Joined with the chain in front of the BPTT launch.
        wg_stream = self._enc_stream if (POSTNET_WGRAD_OVERLAP and getattr(w, "voc_done", None) is not None and not _redo) else None
        w.side_wgrads = wg_stream is not None
        chans = [d.post_ch] * (d.post_n - 1) + [d.n_mel]
        dy = w.d_post
        for i in range(d.post_n - 1, -1, -1):
            cin = d.n_mel if i == 0 else d.post_ch
            x_in = w.linear if i == 0 else w.post_y[i - 1]
            dx = w.post_dx if (i % 2 == 0) else w.post_dx2
            self._conv_block_bwd(dy, x_in, w.post_a[i], w.post_mean[i], w.post_rstd[i], mk["post_drop_%d" % i], 1 - d.conv_drop,
                                 ACT_TANH, "decoder/conv_%d/" % i, B * S, S, cin, chans[i], d.post_k, w.post_dz[i], dx, wgrad_stream=wg_stream)
            dy = dx
        # every postnet gradient is final: its all-reduce overlaps the decoder BPTT - unless that is the persistent launch, which needs
        # every CU of the chip for itself: a collective kernel holding CUs at that moment and the 256 workgroups waiting for each
        # other's CUs would sit out the launch's start window (0.2 s) and end in the fallback.  Then the range is announced BEHIND the
        # launch (the collective is ordered after what is enqueued) and runs under the hoisted weight-gradient products instead.
        postnet_ready_deferred = on_ready is not None and bool(getattr(w, "persist_bwd_now", False)) and bool(getattr(w, "persist_bwd", False)) and not _redo
        if on_ready is not None and not postnet_ready_deferred:
            self._join_vocoder(w)                # (the weight gradients on the side stream are part of the range)
            on_ready(*self._grad_range("decoder/conv_"))
        # d_linear(total) = loss part + residual (d_post) + postnet input grad
        n = B * S * d.n_mel
        call("mstts_add", ptr(w.d_linear), ptr(w.d_post), ptr(w.d_linear), n)
        call("mstts_add", ptr(w.d_linear), ptr(dy), ptr(w.d_linear), n)
        # ---- projection backward
        call("mstts_pack_dproj", ptr(w.d_linear), ptr(w.d_stop), ptr(w.d_proj), self.proj_ld, B, S, d.n_mel)
        self.dwp_pad.zero_()
        self._gemm(w.pj, w.d_proj, self.dwp_pad, H + M, self.proj_ld, S * B, H + M, self.proj_ld, self.proj_ld, trans_a=True,
             split_k=max(2, _split_k(H + M, self.proj_ld, S * B)))
        gwp, ogwp = self.G("decoder/decoder/linear_projection/dense/kernel")
        gbp, ogbp = self.G("decoder/decoder/linear_projection/dense/bias")
        call("mstts_copy2d", ptr(self.dwp_pad), self.proj_ld, ptr(gwp, ogwp), d.n_mel + 1, H + M, d.n_mel + 1, 1)
        call("mstts_colsum", ptr(w.d_proj), S * B, d.n_mel + 1, self.proj_ld, ptr(gbp, ogbp), 1)
        self._gemm(w.d_proj, self.wp_pad, w.d_pj, S * B, H + M, self.proj_ld, self.proj_ld, self.proj_ld, H + M, trans_b=True)
        # ---- decoder loop backward (its persistent launch needs every CU: the vocoder side chain of the forward pass, which ran under the
        # loss and the postnet's backward pass, has to be off the chip)
        self._join_vocoder(w)
        w.dq_hist.zero_()
        db = w.dec_b
        db.fwd = C.pointer(w.dec)
        db.d_pj, db.dg0, db.dg1, db.dq_hist, db.de_hist, db.d_in0, db.ws = (ptr(w.d_pj), ptr(w.dg0), ptr(w.dg1), ptr(w.dq_hist),
                                                                             ptr(w.de_hist), ptr(w.d_in0), ptr(w.dec_bwd_ws))
        SB = S * B
        self.dw0f.zero_()
        w.d_keys.zero_()
        self.d_loc_k.zero_()
        parts = w.d_in0_parts
        # (persist_bwd_now: this step's policy decision, taken in forward(); persist_bwd: the plan's flag, which tests clear between two backward passes)
        use_pbwd = bool(getattr(w, "persist_bwd_now", False)) and bool(getattr(w, "persist_bwd", False)) and bool(getattr(w, "opk_valid", False)) and not _redo
        bwd_done = None
        if getattr(w, "opk_valid", False) and not use_pbwd:
            self.unpack_history(w)            # the persistent forward packed the cell operands; the launch-per-step BPTT reads the histories
        if use_pbwd:
            # ONE launch for the whole BPTT; its status words are read while the hoisted weight-gradient products run (no bubble); the
            # launch-per-step loop is the fallback
            pb = w.pdesc_b
            pb.w0pk, pb.w1pk, pb.wqpk, pb.xch, pb.ctrl = ptr(self.pkb[0]), ptr(self.pkb[1]), ptr(self.pkb[2]), ptr(w.xch_b), ptr(w.pctrl_b)
            pb.stamps = ptr(self.persist_bwd_stamps) if self.persist_bwd_stamps is not None else None
            pb.opk = ptr(w.opk)
            pb.selftest_fail_step = int(self.persist_bwd_selftest)
            pb.near_xcd = int(os.environ.get("MSTTS_PERSIST_NEAR", "1") != "0")
            pb.recurrent_bf16 = int(self.persist_bf16)
            call("mstts_decoder_train_bwd_persistent", C.byref(db), C.byref(pb))
            ev = torch.cuda.Event(enable_timing=self.trace_events)
            ev.record()
            self.bptt_end_event = ev             # (bench.py --gpus N: where the first gradient collective starts relative to this)
            w.pctrl_b_host.copy_(w.pctrl_b, non_blocking=True)
            bwd_done = torch.cuda.Event()
            bwd_done.record()
            parts = 1                        # (on success d_in0 slab 0 holds the complete context gradient; a failed launch re-runs the pass)
        else:
            self._ensure_fallback_packs()
            call("mstts_decoder_train_bwd", C.byref(db))
        if postnet_ready_deferred:
            on_ready(*self._grad_range("decoder/conv_"))
        def decoder_products():
            # hoisted weight gradients of the loop.  (Running them chunk by chunk on a second stream under BPTT was measured: the
            # GEMMs' MFMA traffic slows every latency-bound loop kernel by 25-35 %, 105.4 vs 102.2 ms per step - not kept.)
            self._recurrent_wgrads(w, 0, S, part="products")
            g0, og0 = self.G(CELL % 0 + "kernel")
            call("mstts_copy2d", ptr(self.dw0f), 4 * H, ptr(g0, og0 + Pn * 4 * H), 4 * H, M, 4 * H, 1)
            call("mstts_copy2d", ptr(self.dw0f), 4 * H, ptr(g0, og0 + (Pn + M) * 4 * H), 4 * H, M, 4 * H, 1)
            call("mstts_copy2d", ptr(self.dw0f, M * 4 * H), 4 * H, ptr(g0, og0 + (Pn + 2 * M) * 4 * H), 4 * H, H, 4 * H, 1)
            # prenet backward (d_pre = dg0 . W0[:P]^T was produced per chunk above)
            dcur, dnxt = w.d_pre, w.d_pre2
            for i in range(d.prenet_n - 1, -1, -1):
                cin = d.n_mel if i == 0 else Pn
                x_in = w.frames if i == 0 else w.pre_d[i - 1]
                call("mstts_relu_dropout_bwd", ptr(dcur), ptr(w.pre_d[i]), ptr(mk["prenet_drop_%d" % i]), 1 - d.prenet_drop, ptr(dcur), SB * Pn)
                gk, ogk = self.G("decoder/decoder/prenet_%d/dense/kernel" % i); gb, ogb = self.G("decoder/decoder/prenet_%d/dense/bias" % i)
                self._gemm(x_in, dcur, gk, cin, Pn, SB, cin, Pn, Pn, trans_a=True, split_k=max(2, _split_k(cin, Pn, SB)), c_off=ogk)
                call("mstts_colsum", ptr(dcur), SB, Pn, Pn, ptr(gb, ogb), 1)
                if i > 0:
                    k, ok = self.P("decoder/decoder/prenet_%d/dense/kernel" % i)
                    self._gemm(dcur, k, dnxt, SB, cin, Pn, Pn, Pn, cin, trans_b=True, b_off=ok)
                    dcur, dnxt = dnxt, dcur
            gs = {}
            for field, name in (("conv_k", "attention_convolution_dense_layer/conv1d/kernel"), ("conv_b", "attention_convolution_dense_layer/conv1d/bias"),
                                ("dense_k", "attention_convolution_dense_layer/dense/kernel"), ("score_w", "score_layer/weight_w"), ("score_b", "score_layer/bias_b")):
                t, o = self.G(LSA + name)
                gs[field] = ptr(t, o)
            ls = w.dec.lsa
            call("mstts_lsa_unfold_location_grad", ls.conv_k, ls.conv_b, ls.dense_k, ptr(self.d_loc_k), gs["score_b"],
                 gs["conv_k"], gs["conv_b"], gs["dense_k"], d.att_k, d.att_ch, d.att)

        def memory_gradient():
            # d_values[b] = sum_s align[s,b,:]^T (d_ctx from projection + d_ctx from next step's cell 0)
            self._gemm(w.align_hist, w.d_pj, w.d_values, Te, M, S, B * Te, B * (H + M), M, trans_a=True, batch=B,
                       strides=(Te, H + M, Te * M), b_off=H, exact=True)
            if S > 1:
                for part in range(parts):
                    self._gemm(w.align_hist, w.d_in0, w.d_values, Te, M, S - 1, B * Te, B * (M + H), M, trans_a=True, batch=B,
                               strides=(Te, M + H, Te * M), b_off=(part * S + 1) * B * (M + H), accumulate=True, exact=True)
            # memory layer
            wm, owm = self.P("attention/memory_layer/kernel"); gwm, ogwm = self.G("attention/memory_layer/kernel")
            self._gemm(w.values, w.d_keys, gwm, M, A, B * Te, M, A, A, trans_a=True, split_k=max(2, _split_k(M, A, B * Te)), c_off=ogwm)
            self._gemm(w.d_keys, wm, w.d_values, B * Te, M, A, A, A, M, trans_b=True, accumulate=True, b_off=owm)

        # ---- encoder BiLSTM backward: descriptors
        x_in, cin = w.enc_y[-1], d.enc_conv_ch
        bseqs = []
        for di, dr in enumerate(("fw", "bw")):
            k, ok = self.P(ENC_CELL % dr + "kernel")
            q = lib.LstmSeqBwd()
            q.B, q.T, q.H = B, Te, He
            q.wh = ptr(k, ok + cin * 4 * He); q.wh_ld = 4 * He
            q.lengths = ptr(tlen); q.reverse = di; q.zoneout = d.zoneout
            q.zc = ptr(mk["enc_zc_" + dr]); q.zh = ptr(mk["enc_zh_" + dr])
            q.d_out = ptr(w.d_values, di * He); q.dout_sb = Te * M; q.dout_st = M
            q.c_hist = ptr(w.enc_c[dr]); q.acts = ptr(w.enc_acts[dr]); q.c_raw = ptr(w.enc_craw[dr])
            q.dgates_step = ptr(w.enc_dgs[dr]); q.dgates_pos = ptr(w.enc_dgp[dr]); q.ws = ptr(w.enc_bwd_ws[dr])
            bseqs.append(q)
        def encoder_tail():
            """Everything behind the encoder's BPTT: the BiLSTM's weight gradients and input gradient, the convolution blocks, the embedding."""
            x_in, cin = w.enc_y[-1], d.enc_conv_ch
            for di, dr in enumerate(("fw", "bw")):
                k, ok = self.P(ENC_CELL % dr + "kernel")
                gk, ogk = self.G(ENC_CELL % dr + "kernel"); gb, ogb = self.G(ENC_CELL % dr + "bias")
                self._gemm(x_in, w.enc_dgp[dr], gk, cin, 4 * He, B * Te, cin, 4 * He, 4 * He, trans_a=True,
                     split_k=max(2, _split_k(cin, 4 * He, B * Te)), c_off=ogk)
                self._gemm(w.enc_h[dr], w.enc_dgs[dr], gk, He, 4 * He, Te * B, He, 4 * He, 4 * He, trans_a=True,
                     split_k=max(2, _split_k(He, 4 * He, B * Te)), c_off=ogk + cin * 4 * He)
                call("mstts_colsum", ptr(w.enc_dgs[dr]), Te * B, 4 * He, 4 * He, ptr(gb, ogb), 1)
                self._gemm(w.enc_dgp[dr], k, w.enc_dy, B * Te, cin, 4 * He, 4 * He, 4 * He, cin, trans_b=True, accumulate=(di == 1), b_off=ok)
            # ---- encoder conv backward
            dy = w.enc_dy
            bufs = [w.enc_dx, w.enc_dy]
            for i in range(d.enc_conv_n - 1, -1, -1):
                cin = d.emb if i == 0 else d.enc_conv_ch
                x_in = w.emb if i == 0 else w.enc_y[i - 1]
                dx = bufs[(d.enc_conv_n - 1 - i) % 2]
                self._conv_block_bwd(dy, x_in, w.enc_a[i], w.enc_mean[i], w.enc_rstd[i], mk["enc_conv_drop_%d" % i], 1 - d.conv_drop,
                                     ACT_RELU, "encoder/conv_%d/" % i, B * Te, Te, cin, d.enc_conv_ch, d.enc_conv_k, w.enc_dz, dx)
                dy = dx
            ge, oge = self.G("encoder/embedding_variable")
            call("mstts_embedding_bwd", ptr(tok), ptr(dy), ptr(ge, oge), B * Te, d.n_tok, d.emb)

        # (the persistent BPTT reads the packed history of a persistent forward)
        enc_ticket = None
        enc_persistent = bool(getattr(w, "enc_hist_valid", False)) and not _redo
        if enc_persistent and ENC_OVERLAP:
            # The encoder's persistent BPTT occupies 32 of the 256 CUs for 0.65 ms: it runs on its own stream UNDER the decoder's hoisted weight-gradient
            # products instead of in front of the encoder's.  What it waits for - the attention parameter gradients' d_keys, the memory gradient -
            # is computed first; its launch is enqueued before the products, so its 32 workgroups are resident when the products' tiles arrive.
            self._recurrent_wgrads(w, 0, S, part="attention")
            memory_gradient()
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._enc_stream):
                self._enc_stream.wait_event(ready)
                enc_ticket = self._enc_persistent(w, "mstts_lstm_seq_bwd_pair_persistent", bseqs, 1, 32)
                if ENC_TAIL_OVERLAP:
                    # ... and the rest of the encoder's backward pass (weight gradients, convolution blocks, embedding: ~45 launches, 1.1 ms, none of
                    # it read by the decoder's products) stays on this stream, behind the launch and beside those products
                    encoder_tail()
                enc_done = torch.cuda.Event()
                enc_done.record()
            decoder_products()
            if on_ready is not None:         # decoder + attention gradients are final: overlaps the encoder backward
                on_ready(*self._grad_range("attention/", "decoder/decoder"))
            torch.cuda.current_stream().wait_event(enc_done)
            if not ENC_TAIL_OVERLAP:
                encoder_tail()
        else:
            self._recurrent_wgrads(w, 0, S, part="attention")
            decoder_products()
            memory_gradient()
            if on_ready is not None:
                on_ready(*self._grad_range("attention/", "decoder/decoder"))
            if enc_persistent:
                enc_ticket = self._enc_persistent(w, "mstts_lstm_seq_bwd_pair_persistent", bseqs, 1, 32)
            else:
                self._ensure_fallback_packs()
                call("mstts_lstm_seq_bwd_pair", C.byref(bseqs[0]), C.byref(bseqs[1]))    # BPTT of both directions: two launches per step
            encoder_tail()
        if on_ready is not None:
            on_ready(*self._grad_range("encoder/"))
        # ---- the status words of this pass's persistent launches: ONE host sync (the side stream runs in order)
        job_flag = None
        if agree_async is not None and not _redo:
            # every rank, every pass (also one that launched nothing persistent: its word is 1) - the ranks' collective sequences stay in step
            with torch.cuda.stream(self._side):
                # behind both launches: the decoder's BPTT ended before `bwd_done`, the encoder's before its ticket's event - two waits for
                # events that have (long) fired when the side stream gets here, at the end of the pass
                if bwd_done is not None:
                    self._side.wait_event(bwd_done)
                if enc_ticket is not None:
                    self._side.wait_event(enc_ticket[0])
                if enc_ticket is None and bwd_done is None:      # (nothing persistent in this pass: order the word behind the pass so far)
                    self._side.wait_stream(torch.cuda.current_stream())
                call("mstts_persist_status", ptr(w.pctrl_b) if bwd_done is not None else None, 256,
                     ptr(w.enc_ctrl) if enc_ticket is not None else None, enc_ticket[2] if enc_ticket is not None else 0, ptr(w.job_flag))
                agree_async(w.job_flag)
                w.job_flag_host.copy_(w.job_flag, non_blocking=True)
                job_flag = torch.cuda.Event()
                job_flag.record()
        passed = True
        if enc_ticket is not None:
            passed = self._enc_check(w, enc_ticket)
        if bwd_done is not None:
            bwd_done.synchronize()
            st = w.pctrl_b_host
            if int(st[1]) != 0 or int(st[2]) != 256:
                self.persist_bwd_fallbacks += 1
                self._step_fell_back = True
                self.persist_last_status = (int(st[0]), int(st[1]), int(st[2]))
                passed = False
        if job_flag is not None:
            job_flag.synchronize()
            job_ok = int(w.job_flag_host[0]) == 1            # MIN over the ranks of what mstts_persist_status saw on each
            if passed and not job_ok:
                self.collective_redos += 1
            passed = passed and job_ok
        else:
            passed = self._pass_verdict(passed, agree, _redo)
        if not passed:
            if on_abort is not None:
                on_abort()
            torch.cuda.current_stream().synchronize()
            return self.loss_and_backward(w, grad_scale=grad_scale, on_ready=on_ready, on_abort=on_abort, agree=agree, agree_async=agree_async, _redo=True)

    def _pass_verdict(self, passed, agree, redo):
        """Keep this backward pass or run it again?  With `agree` (data parallel) the answer is the job's, not the rank's: the minimum of
        `passed` over the ranks.  A re-run takes no persistent launch, so it has nothing left to disagree about and asks nobody."""
        if agree is not None and not redo:
            mine = passed
            passed = bool(agree(passed))
            if mine and not passed:
                self.collective_redos += 1
        return passed

    def _recurrent_wgrads(self, w, lo, hi, part="all"):
        """Weight gradients of the decoder loop summed over the steps [lo, hi) (accumulating into the gradient slab):
        dW1, db1, dw0f (folded cell-0 rows), cell-0 prenet rows, db0, dWq, the attention parameter gradients and d_keys; also
        that range of d_pre = dg0 . W0[:P]^T for the prenet backward.  part: "attention" = the attention parameter gradients and d_keys only
        (what the encoder's gradient path waits for), "products" = everything else, "all"."""
        d = self.d
        B, Te = w.B, w.Te
        H, M, A, Pn = d.dec_lstm, d.mem, d.att, d.prenet
        n = (hi - lo) * B
        r = lo * B                                           # first row of the range in the step-major histories
        if part in ("all", "attention"):
            gsw, ogsw = self.G(LSA + "score_layer/weight_w"); gsb, ogsb = self.G(LSA + "score_layer/bias_b")
            call("mstts_lsa_param_bwd", C.byref(w.dec.lsa), hi - lo, ptr(w.q_hist, r * A), ptr(w.cum_hist, r * Te), ptr(w.de_hist, r * Te), ptr(w.d_keys),
                 ptr(self.d_loc_k), ptr(gsw, ogsw), ptr(gsb, ogsb), ptr(w.lsa_param_ws))
        if part == "attention":
            return
        g1, og1 = self.G(CELL % 1 + "kernel"); gb1, ogb1 = self.G(CELL % 1 + "bias")
        g0, og0 = self.G(CELL % 0 + "kernel"); gb0, ogb0 = self.G(CELL % 0 + "bias")
        # Order: the two bias sums (streaming, ~70 us each) first, then the cell-0 product, then the cell-1 product.  The encoder's persistent
        # BPTT (32 workgroups, its own stream) is launched just in front of this: the bias sums give its memsets and workgroups the time to
        # take their 32 CUs, and at the reference widths the folded cell-0 gradient is 7 x 16 tiles x 2 pieces = 224 workgroups of the
        # one-workgroup-per-CU kernel - exactly the CUs that are left - where the cell-1 gradient's 256 would wait a whole round for them
        # (profiles/r06_one_rank_rccl_timeline_before.txt).
        call("mstts_colsum", ptr(w.dg1, r * 4 * H), n, 4 * H, 4 * H, ptr(gb1, ogb1), 1)
        call("mstts_colsum", ptr(w.dg0, r * 4 * H), n, 4 * H, 4 * H, ptr(gb0, ogb0), 1)
        self._gemm(w.in0, w.dg0, self.dw0f, M + H, 4 * H, n, M + H, 4 * H, 4 * H, trans_a=True, split_k=_split_k(M + H, 4 * H, n), accumulate=True,
             a_off=r * (M + H), b_off=r * 4 * H)
        self._gemm(w.in1, w.dg1, g1, 2 * H, 4 * H, n, 2 * H, 4 * H, 4 * H, trans_a=True, split_k=_split_k(2 * H, 4 * H, n), accumulate=True,
             a_off=r * 2 * H, b_off=r * 4 * H, c_off=og1)
        self._gemm(w.pre_d[-1], w.dg0, g0, Pn, 4 * H, n, Pn, 4 * H, 4 * H, trans_a=True, split_k=max(2, _split_k(Pn, 4 * H, n)),
             a_off=r * Pn, b_off=r * 4 * H, c_off=og0)
        k0, o0 = self.P(CELL % 0 + "kernel")
        self._gemm(w.dg0, k0, w.d_pre, n, Pn, 4 * H, 4 * H, 4 * H, Pn, trans_b=True, a_off=r * 4 * H, b_off=o0, c_off=r * Pn)
        gq, ogq = self.G(LSA + "query_layer/kernel")
        self._gemm(w.pj, w.dq_hist, gq, H, A, n, H + M, A, A, trans_a=True, split_k=max(2, _split_k(H, A, n)), a_off=r * (H + M), b_off=r * A, c_off=ogq)

    def _grad_range(self, *prefixes):
        """[lo, hi) of the gradient slab covered by the trainable variables whose names start with one of `prefixes` (the
        variable table keeps each module contiguous: encoder | attention | decoder | postnet)."""
        ps = self.params
        offs = [(ps.offset[n], ps.offset[n] + (int(np.prod(ps.shape[n])) + 3) // 4 * 4) for n, _, _ in ps.table
                if ps.trainable[n] and n.startswith(prefixes)]
        return min(o for o, _ in offs), max(e for _, e in offs)

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, grad_scale=1.0):
        """tf.train.AdamOptimizer + the 1e-6 * l2 regulariser's gradient (MSTTS_SV.py:145-176)."""
        ps = self.params
        b1, b2, eps = self.adam
        t = self.global_step + 1
        lr = learning_rate(self.global_step)
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        call("mstts_adam_tf", ptr(ps.train), ptr(ps.grad), ptr(ps.adam_m), ptr(ps.adam_v), ptr(ps.wd_mask), float(self.wr_rate),
             float(grad_scale), float(lr_t), b1, b2, eps, ps.n_train)
        self.global_step += 1
        ps.touch()
        self._derived_stale = True
        self.refresh_derived()
        return lr

    def moving_stat_ranges(self):
        """Merged [lo, hi) ranges of the non-trainable slab that hold batch-norm moving statistics (the only per-rank state of a
        data-parallel run besides the loss scalars)."""
        ps = self.params
        r = sorted((ps.offset[n], ps.offset[n] + (int(np.prod(ps.shape[n])) + 3) // 4 * 4) for n, _, _ in ps.table
                   if not ps.trainable[n] and n.endswith(("moving_mean", "moving_variance")))
        out = []
        for lo, hi in r:
            if out and out[-1][1] == lo:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
        return [tuple(x) for x in out]

    def sync_statistics(self, group=None):
        """Average the BN moving statistics over the ranks (batch statistics stay per rank - each rank is the reference
        at batch 32 - but what gets reported / checkpointed must not depend on which rank writes it)."""
        from .dist import average_
        average_([self.params.frozen[lo:hi] for lo, hi in self.moving_stat_ranges()], group=group)

    def scalars(self, w, average=False, group=None):
        """Loss scalars of the last step on `w`; average=True: mean over the data-parallel ranks."""
        if average:
            from .dist import average_
            avg = w.scalars.clone()
            average_([avg], group=group)
            s = avg.cpu().numpy()
        else:
            s = w.scalars.detach().cpu().numpy()
        wr = float(s[3]) * self.wr_rate
        return {"Linear_Loss": float(s[0]), "Postnet_Loss": float(s[1]), "Stop_Loss": float(s[2]),
                "Weight_Regularization_Loss": wr, "Loss": float(s[0] + s[1] + s[2]) + wr}

    def scalars_async(self, w, average=False, group=None):
        """The same, without waiting for the step: the four loss words are copied to a page-locked slot (four slots in rotation) behind
        everything enqueued so far and a handle is returned; handle.get() waits for THAT copy only.  Tacotron2.Train reads a step's losses
        while the next step runs, so the host never drains the device between two steps (MSTTS_SV.py:270-273 fetches them with the step)."""
        if getattr(self, "_scalar_ring", None) is None:
            self._scalar_ring = torch.zeros(4, 4, dtype=torch.float32).pin_memory()
            self._scalar_next = 0
            self._scalar_avg = torch.zeros(4, dtype=torch.float32, device=self.device)
            self._scalar_handles = [None] * 4
        old = self._scalar_handles[self._scalar_next]
        if old is not None:
            old.get()                        # a handle nobody has read yet takes its words out of the slot before the slot is used again (its copy ended steps ago)
        slot = self._scalar_ring[self._scalar_next]
        src = w.scalars
        if average:
            from .dist import average_
            self._scalar_avg.copy_(w.scalars)
            average_([self._scalar_avg], group=group)
            src = self._scalar_avg
        slot.copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        h = self._scalar_handles[self._scalar_next] = _LateScalars(ev, slot, self.wr_rate)
        self._scalar_next = (self._scalar_next + 1) % 4
        return h

    def exchange_timeouts(self, w):
        """Count of in-launch exchange time-outs of the single-launch forward attention kernel in the last step on workspace `w`
        (csrc/lsa.hip: the word behind the last granule).  Always 0 on a healthy run; non-zero means a workgroup fell back to its
        serial recompute.  (The backward kernel needs no exchange.)  Synchronises."""
        return int(w.energy_ws.view(torch.int64)[w.B * w.Te].item())

    def broadcast_state(self, src=0, group=None):
        """Data-parallel start: every rank takes rank `src`'s variables, Adam slots and statistics."""
        from .dist import broadcast_
        ps = self.params
        step = torch.tensor([self.global_step], dtype=torch.int64, device=self.device)
        broadcast_([ps.train, ps.frozen, ps.adam_m, ps.adam_v, step], src=src, group=group)
        self.global_step = int(step.item())
        ps.touch()
        self._derived_stale = True

    def train_step(self, batch, masks=None, all_reduce=None):
        """One full iteration: forward, loss, backward, (gradient all-reduce), Adam."""
        if self.deterministic and not getattr(lib.deterministic_gemm._depth, "n", 0):
            with lib.deterministic_gemm():
                return self.train_step(batch, masks=masks, all_reduce=all_reduce)
        B, Te = batch["Token"].shape
        L = batch["Mel"].shape[1]
        w = self.plan(B, Te, L)
        self.forward(batch, w, masks=masks, allowed=self._persist_begin_step(), overlap_vocoder=True)      # the fallback policy advances once per optimizer step
        if all_reduce is not None:           # bucketed in the order gradients become final (postnet -> decoder/attention -> encoder),
            g = self.params.grad             # each bucket's all-reduce running under the rest of the backward pass
            self.loss_and_backward(w, on_ready=lambda lo, hi: all_reduce.start(g, lo, hi), on_abort=lambda: all_reduce.finish(g),
                                   agree=all_reduce.agree, agree_async=all_reduce.agree_async if getattr(all_reduce, "async_flag", False) else None)
            all_reduce.finish(g)
        else:
            self.loss_and_backward(w)
        self.adam_step(grad_scale=1.0 / self.world if all_reduce is not None else 1.0)
        return w
