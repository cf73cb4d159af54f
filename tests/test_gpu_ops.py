"""Op-level parity of the HIP kernels (called through the C ABI) against NumPy / the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from multi_speaker_tts_amd import lib
from oracle import rng as orng
from tests.helpers import rel_err, t2n

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _r(dev, *shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.tensor(g.normal(0, scale, size=shape), dtype=torch.float32, device=dev)


@pytest.fixture(params=["split3", "f32_mfma"])
def gemm_mode(request):
    """The fp32 contractions' two inner products: the six-product bf16 split (default, csrc/gemm_split.inc) and v_mfma_f32_32x32x2_f32."""
    lib.call("mstts_gemm_split3", 1 if request.param == "split3" else 0)
    yield request.param
    lib.call("mstts_gemm_split3", 1)


@pytest.mark.parametrize("M,N,K,ta,win,sk", [(1792, 4096, 25632, 1, None, 4), (4096, 512, 2560, 0, (128, 512, 2), 1), (2560, 512, 6408, 1, (801, 512, 2), 3),
                                              (25632, 256, 4096, 0, None, 1)])
def test_gemm_split_is_fp32_accurate(dev, M, N, K, ta, win, sk):
    """The six-product bf16 split against fp64 on the step's own big shapes (weight gradients over 25 632 rows, the 512-channel convolutions,
    the prenet data gradient): its error must not exceed the error of the exact-fp32 matrix-core kernel on the same operands (both are fp32
    accumulations of - to 2^-26 - the same products; measured: equal to within the run-to-run spread of the split-K atomics), and both sit at
    the fp32 level (1e-6 of the result's scale), three orders of magnitude below one-term bf16 (test_gemm_bf16_layouts: 1e-2)."""
    g = np.random.default_rng(5)
    if win:
        T, cin, pad = win
        rows = M if not ta else K
        x = torch.tensor(g.normal(0, 1, (rows, cin)), dtype=torch.float32, device=dev)
        Kt = (K if not ta else M) // cin
        xp = t2n(x).astype(np.float64).reshape(rows // T, T, cin)
        xp = np.pad(xp, ((0, 0), (pad, Kt - 1 - pad), (0, 0)))
        wins = np.stack([xp[:, k:k + T] for k in range(Kt)], axis=2).reshape(rows, Kt * cin)
        a64 = wins.T if ta else wins
        A = x
    else:
        A = torch.tensor(g.normal(0, 1, (K, M) if ta else (M, K)), dtype=torch.float32, device=dev)
        a64 = t2n(A).astype(np.float64)
        a64 = a64.T if ta else a64
    B = torch.tensor(g.normal(0, 1, (K, N)), dtype=torch.float32, device=dev)
    ref = a64 @ t2n(B).astype(np.float64)
    errs = {}
    for mode in (1, 0):
        lib.call("mstts_gemm_split3", mode)
        Cm = torch.zeros(M, N, device=dev)
        lib.gemm(A, B, Cm, M, N, K, A.shape[1], N, N, trans_a=bool(ta), win=win, split_k=sk)
        errs[mode] = rel_err(t2n(Cm), ref)
    lib.call("mstts_gemm_split3", 1)
    print("max error / max |ref|: split %.3e, f32 MFMA %.3e" % (errs[1], errs[0]))
    assert errs[0] < 5e-6 and errs[1] < 5e-6
    assert errs[1] <= 1.25 * errs[0] + 1e-8, errs


@pytest.mark.parametrize("M,N,K,ta,tb,sk", [(2048, 4096, 2570, 1, 0, 2), (6000, 2200, 520, 0, 0, 1), (3584, 4100, 333, 0, 1, 1), (2050, 3330, 96, 1, 1, 3)])
def test_gemm_split_big_tile(dev, M, N, K, ta, tb, sk):
    """The 256 x 256 x 16 form of the six-product split (round 5; taken when the output fills the chip with such tiles): every layout, ragged
    edges in M, N and K, unaligned (scalar-load) shapes, split-K atomics onto a pre-filled output, bias + activation - against fp64 at the
    fp32 level, and equal to the 128 x 128 x 32 producer / consumer kernel to fp32 summation order."""
    A = _r(dev, *((K, M) if ta else (M, K)), seed=1)
    B = _r(dev, *((N, K) if tb else (K, N)), seed=2, scale=1.0 / np.sqrt(K))
    bias = _r(dev, N, seed=3)
    a = t2n(A).astype(np.float64); b = t2n(B).astype(np.float64)
    prod = (a.T if ta else a) @ (b.T if tb else b)
    outs = {}
    for big in (1, 0):
        lib.load().mstts_gemm_split_big(big)
        if sk > 1:
            Cm = torch.ones(M, N, device=dev)
            _gemm_call("mstts_gemm_f32", A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, split_k=sk, bias=bias)
            ref = 1.0 + prod + t2n(bias).astype(np.float64)
        else:
            Cm = torch.zeros(M, N, device=dev)
            _gemm_call("mstts_gemm_f32", A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, bias=bias, act=2)
            ref = np.tanh(prod + t2n(bias).astype(np.float64))
        outs[big] = t2n(Cm).astype(np.float64)
        assert rel_err(outs[big], ref) < 5e-6, (big, rel_err(outs[big], ref))
    lib.load().mstts_gemm_split_big(1)
    assert rel_err(outs[1], outs[0]) < 5e-6


def test_gemm_split_big_tile_conv_window(dev):
    """... and its implicit-im2col window mode at the postnet's shape class, forward and weight gradient, against fp64."""
    B_, T, cin, cout, K = 40, 801, 64, 512, 5
    x = _r(dev, B_ * T, cin, seed=4); w = _r(dev, K * cin, cout, seed=5, scale=0.1)
    xb = x.double().cpu().reshape(B_, T, cin); wb = w.double().cpu().reshape(K, cin, cout)
    ref = torch.nn.functional.conv1d(xb.transpose(1, 2), wb.permute(2, 1, 0), padding=(K - 1) // 2).transpose(1, 2).reshape(B_ * T, cout).numpy()
    dy = _r(dev, B_ * T, cout, seed=6)
    xpad = torch.nn.functional.pad(xb, (0, 0, (K - 1) // 2, K - 1 - (K - 1) // 2))
    winm = xpad.unfold(1, K, 1).permute(0, 1, 3, 2).reshape(B_ * T, K * cin)
    refw = (winm.t() @ dy.double().cpu()).numpy()
    for big in (1, 0):
        lib.load().mstts_gemm_split_big(big)
        y = torch.zeros(B_ * T, cout, device=dev)
        _gemm_call("mstts_gemm_f32", x, w, y, B_ * T, cout, K * cin, cin, cout, cout, win=(T, cin, (K - 1) // 2))
        assert rel_err(t2n(y), ref) < 5e-6, big
    lib.load().mstts_gemm_split_big(1)
    dw = torch.zeros(K * cin, cout, device=dev)
    _gemm_call("mstts_gemm_f32", x, dy, dw, K * cin, cout, B_ * T, cin, cout, cout, ta=1, win=(T, cin, (K - 1) // 2), split_k=64)
    assert rel_err(t2n(dw), refw) < 5e-6


def _gemm_both(dev, A, B, M, N, K):
    out = {}
    for mode in (1, 0):
        lib.call("mstts_gemm_split3", mode)
        Cm = torch.zeros(M, N, device=dev)
        lib.gemm(A, B, Cm, M, N, K, K, N, N)
        out[mode] = t2n(Cm).astype(np.float64)
    lib.call("mstts_gemm_split3", 1)
    return out[1], out[0]


def test_gemm_split_edge_semantics(dev):
    """What the six-product split (mstts_gemm_split3(1), the default) does at the edges of fp32, next to v_mfma_f32_32x32x2_f32 (Modules.py:29-35:
    tf.layers.conv1d is an IEEE fp32 contraction).  The statements here are the ones include/mstts.h makes next to mstts_gemm_split3:
      (1) finite operands of any magnitude mix (1e-30 ... 1e30 inside one reduction): same error against fp64 as the f32-input MFMA kernel;
      (2) an operand that is +-inf / NaN - or finite but rounding to bf16 infinity, |x| >= 3.3961e38 - makes every output element of its row
          (A) / column (B) NaN, where the f32-input MFMA gives +-inf or NaN: the SET of non-finite outputs is the same, everything else untouched;
      (3) fp32 denormals and the lower planes of operands below ~1e-33 may be flushed by the bf16 matrix cores: absolute error at most
          2^-8 |x| |b| per such product, i.e. invisible unless an output is made of such products only."""
    g = np.random.default_rng(11)
    M, N, K = 256, 256, 512
    # (1) 60 decades of operand magnitude inside every dot product, products O(1)
    e = g.uniform(-30, 30, K)
    A = torch.tensor(g.normal(0, 1, (M, K)) * 10.0 ** e[None, :], dtype=torch.float32, device=dev)
    B = torch.tensor(g.normal(0, 1, (K, N)) * 10.0 ** -e[:, None], dtype=torch.float32, device=dev)
    ref = t2n(A).astype(np.float64) @ t2n(B).astype(np.float64)
    sp, f3 = _gemm_both(dev, A, B, M, N, K)
    e_sp, e_f3 = rel_err(sp, ref), rel_err(f3, ref)
    print("wide range: split %.3e, f32 MFMA %.3e" % (e_sp, e_f3))
    assert np.isfinite(sp).all() and e_f3 < 5e-6 and e_sp <= 1.25 * e_f3 + 1e-7, (e_sp, e_f3)
    # (2) non-finite operands
    A = torch.tensor(g.normal(0, 1, (M, K)), dtype=torch.float32, device=dev)
    B = torch.tensor(g.normal(0, 1, (K, N)), dtype=torch.float32, device=dev)
    B[500] = B[500].clamp(-0.9, 0.9)                                                     # (3.4e38 x 0.9 stays finite in fp32)
    ref = t2n(A).astype(np.float64) @ t2n(B).astype(np.float64)                          # (only the rows / columns NOT touched below are compared with it)
    A[3, 7] = float("inf"); A[5, 9] = float("nan"); A[200, 500] = 3.4e38                 # finite, rounds to bf16 infinity
    B[11, 2] = float("-inf"); B[300, 130] = float("nan")
    sp, f3 = _gemm_both(dev, A, B, M, N, K)
    bad = np.zeros((M, N), bool)
    bad[[3, 5], :] = True; bad[:, [2, 130]] = True
    assert not np.isfinite(f3[bad]).any() and np.isfinite(f3[200][~bad[200]]).all()     # the f32-input MFMA: inf / NaN where IEEE says so; the 3.4e38 row is finite
    assert np.isnan(sp[bad]).all() and np.isnan(sp[200]).all()                          # the split: NaN on all of them, and on the bf16-overflow row
    ok = ~bad; ok[200] = False
    assert np.isfinite(sp[ok]).all() and rel_err(sp[ok], ref[ok]) < 5e-6 and rel_err(f3[ok], ref[ok]) < 5e-6
    # (3) denormal operands inside ordinary rows, and rows made of tiny operands only
    A = torch.tensor(g.normal(0, 1, (M, K)), dtype=torch.float32, device=dev)
    A[:, ::7] *= 1e-40                                                                   # fp32 denormals
    A[64:128] *= 1e-36                                                                   # rows of tiny normals (and denormals): every product tiny
    B = torch.tensor(g.normal(0, 1, (K, N)), dtype=torch.float32, device=dev)
    a64, b64 = t2n(A).astype(np.float64), t2n(B).astype(np.float64)
    ref = a64 @ b64
    sp, f3 = _gemm_both(dev, A, B, M, N, K)
    big = np.ones(M, bool); big[64:128] = False
    e_sp, e_f3 = rel_err(sp[big], ref[big]), rel_err(f3[big], ref[big])
    tiny = lambda x: float(np.abs(x[~big] - ref[~big]).max() / np.abs(ref[~big]).max())       # (rel_err's 1e-12 floor would swallow results of 1e-35)
    t_sp, t_f3 = tiny(sp), tiny(f3)
    print("denormals inside ordinary rows: split %.3e, f32 MFMA %.3e; rows of tiny operands only: split %.3e, f32 MFMA %.3e" % (e_sp, e_f3, t_sp, t_f3))
    assert e_sp < 5e-6 and e_f3 < 5e-6
    assert t_sp <= 2.0 ** -8, t_sp


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (130, 70, 36), (32, 512, 260), (17, 81, 100), (300, 84, 64), (1, 1, 52), (260, 1100, 72)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(dev, gemm_mode, M, N, K, ta, tb):
    A = _r(dev, K, M, seed=1) if ta else _r(dev, M, K, seed=1)
    B = _r(dev, N, K, seed=2) if tb else _r(dev, K, N, seed=2)
    bias = _r(dev, N, seed=3)
    Cm = torch.zeros(M, N, device=dev)
    lib.gemm(A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, bias=bias, trans_a=ta, trans_b=tb, act=lib.ACT_TANH)
    a = t2n(A).astype(np.float64); b = t2n(B).astype(np.float64)
    ref = np.tanh((a.T if ta else a) @ (b.T if tb else b) + t2n(bias))
    assert rel_err(t2n(Cm), ref) < TOL


def test_gemm_splitk_batch_accumulate(dev, gemm_mode):
    M, N, K, nb = 96, 160, 1000, 3
    A = _r(dev, nb, K, M, seed=4); B = _r(dev, nb, K, N, seed=5)
    Cm = torch.ones(nb, M, N, device=dev)
    lib.gemm(A, B, Cm, M, N, K, M, N, N, trans_a=True, split_k=4, batch=nb, strides=(K * M, K * N, M * N))
    ref = 1.0 + np.einsum("bkm,bkn->bmn", t2n(A).astype(np.float64), t2n(B).astype(np.float64))
    assert rel_err(t2n(Cm), ref) < TOL
    Cm2 = torch.full((M, N), 2.0, device=dev)
    lib.gemm(A, B, Cm2, M, N, K, M, N, N, trans_a=True, accumulate=True, alpha=0.5)
    ref2 = 2.0 + 0.5 * (t2n(A[0]).astype(np.float64).T @ t2n(B[0]).astype(np.float64))
    assert rel_err(t2n(Cm2), ref2) < TOL


def test_gemm_dw_shapes_full_tiles(dev, gemm_mode):
    """dW-shaped products (A as [K, M], B as [K, N]) large enough for the 128-row tile (the small cases above all take 64-row tiles):
    plain with ragged M / N / K and split-K onto ones, and a conv weight gradient (window)."""
    if True:
        M, N, K = 4000, 1000, 333
        A = _r(dev, K, M, seed=21); B = _r(dev, K, N, seed=22, scale=1.0 / np.sqrt(K))
        for sk in (1, 3):
            Cm = torch.ones(M, N, device=dev)
            lib.gemm(A, B, Cm, M, N, K, M, N, N, trans_a=True, split_k=sk, accumulate=(sk == 1), alpha=0.5)
            ref = 1.0 + 0.5 * (t2n(A).astype(np.float64).T @ t2n(B).astype(np.float64))
            assert rel_err(t2n(Cm), ref) < TOL, sk
        Bn, T, cin, cout, Kt = 3, 37, 800, 1000, 5
        x = _r(dev, Bn, T, cin, seed=23); dy = _r(dev, Bn, T, cout, seed=24, scale=0.1)
        dw = torch.zeros(Kt, cin, cout, device=dev)
        pad = (Kt - 1) // 2
        lib.gemm(x, dy, dw, Kt * cin, cout, Bn * T, cin, cout, cout, trans_a=True, win=(T, cin, pad), split_k=2)
        xp = np.pad(t2n(x).astype(np.float64), ((0, 0), (pad, Kt - 1 - pad), (0, 0)))
        win = np.stack([xp[:, k:k + T] for k in range(Kt)], axis=2).reshape(Bn, T, Kt * cin)
        ref_dw = np.einsum("btk,bto->ko", win, t2n(dy).astype(np.float64)).reshape(Kt, cin, cout)
        assert rel_err(t2n(dw), ref_dw) < TOL


@pytest.mark.parametrize("M,N,K,win,accumulate,act", [(8990, 512, 640, None, False, 0), (8990, 512, 330, None, True, 0), (4 * 2237, 500, 5 * 64, (2237, 64, 2), False, 0),
                                                       (17000, 140, 2048, None, False, 0), (8990, 512, 640, None, False, 2), (4 * 2237, 500, 5 * 64, (2237, 64, 2), False, 1),
                                                       (3000, 500, 2048, None, False, 2), (3000, 500, 2048, None, True, 0), (8 * 128, 512, 5 * 512, (128, 512, 2), False, 1)])   # at most 128 tiles: the split kernel cuts every tile
def test_gemm_body_tail_split(dev, gemm_mode, M, N, K, win, accumulate, act):
    """Tile lists that end in a small fraction of a round (here 284 / 281 / 284 / 266 tiles): the last tiles are cut along K into pieces
    accumulated with atomics onto cleared tile rows - same product as with the split switched off, ragged edges included."""
    A = _r(dev, M, win[1] if win else K, seed=11)
    B = _r(dev, K, N, seed=12, scale=1.0 / np.sqrt(K))
    bias = _r(dev, N, seed=13)
    outs = []
    for on in (1, 0):
        lib.call("mstts_gemm_tail_split", on)
        Cm = torch.full((M + 3, N), 0.5, device=dev)          # three guard rows behind the operand: the clearing must stop at M
        lib.gemm(A, B, Cm, M, N, K, A.shape[1], N, N, bias=bias, win=win, accumulate=accumulate, act=act)
        outs.append(t2n(Cm))
    lib.call("mstts_gemm_tail_split", 1)
    assert np.all(outs[0][M:] == 0.5) and np.all(outs[1][M:] == 0.5)
    assert rel_err(outs[0][:M], outs[1][:M].astype(np.float64)) < 1e-5         # (the two schedules add the K range in different orders)
    if win is None:
        ref = t2n(A).astype(np.float64) @ t2n(B).astype(np.float64) + t2n(bias) + (0.5 if accumulate else 0.0)
        ref = np.tanh(ref) if act == 2 else ref
        assert rel_err(outs[0][:M], ref) < TOL


def test_gemm_deterministic_switch(dev):
    """mstts_gemm_deterministic (per thread): with it a short tile list (96 tiles - cut along K otherwise) and a list with a tail are bit-equal
    from run to run; the context manager nests and restores."""
    for M, N, K in ((3000, 500, 4096), (8990, 512, 640)):
        A = _r(dev, M, K, seed=31); B = _r(dev, K, N, seed=32)
        outs = []
        with lib.deterministic_gemm():
            with lib.deterministic_gemm():
                pass
            for _ in range(3):
                Cm = torch.zeros(M, N, device=dev)
                lib.gemm(A, B, Cm, M, N, K, K, N, N)
                outs.append(t2n(Cm))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        Cm = torch.zeros(M, N, device=dev)
        lib.gemm(A, B, Cm, M, N, K, K, N, N)                 # outside: the K-cut schedules, same product to fp32 rounding
        assert rel_err(t2n(Cm), outs[0].astype(np.float64)) < 1e-5


@pytest.mark.parametrize("K,cin,cout,T", [(5, 32, 48, 19), (1, 8, 16, 7), (2, 8, 12, 9), (8, 8, 20, 33), (3, 64, 8, 140)])
def test_conv1d_same_fwd_bwd(dev, gemm_mode, K, cin, cout, T):
    """conv1d 'same' as windowed GEMM: forward, weight gradient, data gradient vs torch-free NumPy."""
    Bn = 3
    x = _r(dev, Bn, T, cin, seed=6); w = _r(dev, K, cin, cout, seed=7, scale=0.3); bias = _r(dev, cout, seed=8)
    y = torch.zeros(Bn, T, cout, device=dev)
    pad = (K - 1) // 2
    lib.gemm(x, w, y, Bn * T, cout, K * cin, cin, cout, cout, bias=bias, win=(T, cin, pad))
    xn, wn = t2n(x).astype(np.float64), t2n(w).astype(np.float64)
    xp = np.pad(xn, ((0, 0), (pad, K - 1 - pad), (0, 0)))
    win = np.stack([xp[:, k:k + T] for k in range(K)], axis=2).reshape(Bn, T, K * cin)
    ref = win @ wn.reshape(K * cin, cout) + t2n(bias)
    assert rel_err(t2n(y), ref) < TOL
    dy = _r(dev, Bn, T, cout, seed=9)
    dw = torch.zeros(K, cin, cout, device=dev)
    lib.gemm(x, dy, dw, K * cin, cout, Bn * T, cin, cout, cout, trans_a=True, win=(T, cin, pad), split_k=2)
    dyn = t2n(dy).astype(np.float64)
    ref_dw = np.einsum("btk,bto->ko", win, dyn).reshape(K, cin, cout)
    assert rel_err(t2n(dw), ref_dw) < TOL
    wt = torch.zeros(K, cout, cin, device=dev)
    lib.call("mstts_conv_kernel_flip", lib.ptr(w), lib.ptr(wt), K, cin, cout)
    dx = torch.zeros(Bn, T, cin, device=dev)
    lib.gemm(dy, wt, dx, Bn * T, cin, K * cout, cout, cin, cin, win=(T, cout, K - 1 - pad))
    ref_dx = np.zeros_like(xp)
    dwin = (dyn @ wn.reshape(K * cin, cout).T).reshape(Bn, T, K, cin)
    for k in range(K):
        ref_dx[:, k:k + T] += dwin[:, :, k]
    ref_dx = ref_dx[:, pad:pad + T]
    assert rel_err(t2n(dx), ref_dx) < TOL


def test_philox_bit_exact(dev):
    for n, seed, stream, keep in [(1000, 1234, 7, 0.5), (4099, 2 ** 40 + 17, 1031, 0.9), (3, 5, 0, 0.1)]:
        out = torch.zeros((n + 3) // 4 * 4, dtype=torch.uint8, device=dev)
        lib.call("mstts_philox_keep_mask", lib.ptr(out), n, seed, stream, keep)
        assert np.array_equal(t2n(out)[:n], orng.keep_mask((n,), seed, stream, keep))


def test_embedding_bit_exact(dev):
    tab = _r(dev, 42, 64, seed=1)
    tok = torch.tensor(np.random.default_rng(2).integers(0, 42, size=(5, 9)), dtype=torch.int32, device=dev)
    out = torch.zeros(45, 64, device=dev)
    lib.call("mstts_embedding_fwd", lib.ptr(tok), lib.ptr(tab), lib.ptr(out), 45, 42, 64)
    assert np.array_equal(t2n(out), t2n(tab)[t2n(tok).reshape(-1)])
    dt = torch.zeros(42, 64, device=dev)
    dout = _r(dev, 45, 64, seed=3)
    lib.call("mstts_embedding_bwd", lib.ptr(tok), lib.ptr(dout), lib.ptr(dt), 45, 42, 64)
    ref = np.zeros((42, 64)); np.add.at(ref, t2n(tok).reshape(-1), t2n(dout).astype(np.float64))
    assert rel_err(t2n(dt), ref) < TOL


@pytest.mark.parametrize("act", [lib.ACT_RELU, lib.ACT_TANH])
def test_bn_dropout_fwd_bwd(dev, act):
    rows, Cc = 333, 48
    z = _r(dev, rows, Cc, seed=1)
    x = torch.relu(z) if act == lib.ACT_RELU else torch.tanh(z)
    gamma, beta = _r(dev, Cc, seed=2) + 1.5, _r(dev, Cc, seed=3)
    mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    mask = torch.tensor(orng.keep_mask((rows, Cc), 9, 1, 0.5), device=dev)
    y = torch.zeros_like(x); sm = torch.zeros(Cc, device=dev); sr = torch.zeros(Cc, device=dev); ws = torch.zeros(2 * Cc, device=dev)
    lib.call("mstts_bn_train_fwd", lib.ptr(x), lib.ptr(gamma), lib.ptr(beta), lib.ptr(mm), lib.ptr(mv), lib.ptr(y), lib.ptr(sm), lib.ptr(sr),
             lib.ptr(mask), 0.5, 0.99, 1e-3, rows, Cc, lib.ptr(ws))
    xd = x.double().cpu().requires_grad_(True); g = gamma.double().cpu().requires_grad_(True); b = beta.double().cpu().requires_grad_(True)
    mean = xd.mean(0); var = ((xd - mean) ** 2).mean(0)
    yr = ((xd - mean) / torch.sqrt(var + 1e-3) * g + b) * mask.cpu().double() / 0.5
    assert rel_err(t2n(y), t2n(yr)) < TOL
    assert rel_err(t2n(mm), 0.01 * t2n(mean)) < TOL and rel_err(t2n(mv), 0.99 + 0.01 * t2n(var)) < TOL
    dy = _r(dev, rows, Cc, seed=5)
    # reference gradient wrt the conv pre-activation z through act -> BN -> dropout
    zd = z.double().cpu().requires_grad_(True)
    xa = torch.relu(zd) if act == lib.ACT_RELU else torch.tanh(zd)
    mean = xa.mean(0); var = ((xa - mean) ** 2).mean(0)
    yr = ((xa - mean) / torch.sqrt(var + 1e-3) * g + b) * mask.cpu().double() / 0.5
    (yr * dy.double().cpu()).sum().backward()
    dz = torch.zeros_like(x); dg = torch.zeros(Cc, device=dev); db = torch.zeros(Cc, device=dev); dbias = torch.zeros(Cc, device=dev)
    lib.call("mstts_bn_train_bwd", lib.ptr(dy), lib.ptr(x), lib.ptr(gamma), lib.ptr(sm), lib.ptr(sr), lib.ptr(mask), 0.5, act, lib.ptr(dz),
             lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), rows, Cc, lib.ptr(ws))
    assert rel_err(t2n(dz), t2n(zd.grad)) < 5e-5
    assert rel_err(t2n(dg), t2n(g.grad)) < 5e-5 and rel_err(t2n(db), t2n(b.grad)) < 5e-5
    assert rel_err(t2n(dbias), t2n(zd.grad.sum(0))) < 1e-4


def test_misc_elementwise(dev):
    x = _r(dev, 4, 9, 12, seed=1)
    y = torch.zeros_like(x)
    lib.call("mstts_maxpool2_same", lib.ptr(x), lib.ptr(y), 4, 9, 12)
    xn = t2n(x); ref = np.maximum(xn, np.concatenate([xn[:, 1:], np.full_like(xn[:, :1], -np.inf)], 1))
    assert np.array_equal(t2n(y), ref)
    src = _r(dev, 10, 7, seed=2)
    s = t2n(src); ref = np.concatenate([s[:2], s[2:5] + s[5:8], s[8:]], 0)
    dst = torch.zeros(7, 7, device=dev)
    lib.call("mstts_fold_rows", lib.ptr(src), lib.ptr(dst), 10, 7, 2, 3)
    assert np.allclose(t2n(dst), ref)
    mel = _r(dev, 3, 5, 8, seed=3); fr = torch.zeros(6, 3, 8, device=dev)
    lib.call("mstts_shift_frames", lib.ptr(mel), lib.ptr(fr), 3, 5, 8)
    ref = np.concatenate([np.zeros((1, 3, 8), np.float32), t2n(mel).transpose(1, 0, 2)], 0)
    assert np.array_equal(t2n(fr), ref)
    cs = torch.zeros(7, device=dev)
    lib.call("mstts_colsum", lib.ptr(src), 10, 7, 7, lib.ptr(cs), 0)
    assert rel_err(t2n(cs), s.astype(np.float64).sum(0)) < TOL


def test_adam_tf(dev):
    n = 1001
    p, g = _r(dev, n, seed=1), _r(dev, n, seed=2)
    m, v = _r(dev, n, seed=3) * 0.1, torch.abs(_r(dev, n, seed=4)) * 0.01
    wd = torch.tensor(np.random.default_rng(5).integers(0, 2, n).astype(np.uint8), device=dev)
    pn, gn, mn, vn = [t2n(t).astype(np.float64) for t in (p, g, m, v)]
    lib.call("mstts_adam_tf", lib.ptr(p), lib.ptr(g), lib.ptr(m), lib.ptr(v), lib.ptr(wd), 1e-6, 0.5, 3e-4, 0.9, 0.999, 1e-6, n)
    gt = gn * 0.5 + 1e-6 * pn * t2n(wd)
    mr = 0.9 * mn + 0.1 * gt; vr = 0.999 * vn + 0.001 * gt * gt
    assert rel_err(t2n(p), pn - 3e-4 * mr / (np.sqrt(vr) + 1e-6)) < 1e-6
    assert rel_err(t2n(m), mr) < 1e-6 and rel_err(t2n(v), vr) < 1e-6


def _gemm_call(name, A, B, Cm, M, N, K, lda, ldb, ldc, **kw):
    d = lib.GemmDesc()
    d.A, d.B, d.C, d.bias = lib.ptr(A), lib.ptr(B), lib.ptr(Cm), lib.ptr(kw.get("bias"))
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.trans_a, d.trans_b = int(kw.get("ta", 0)), int(kw.get("tb", 0))
    if kw.get("win"):
        d.win_T, d.win_C, d.win_pad = kw["win"]
        d.win_dil = 1
    d.act, d.accumulate, d.split_k = kw.get("act", 0), int(kw.get("accumulate", 0)), kw.get("split_k", 1)
    d.batch, d.alpha = 1, 1.0
    lib.call(name, C.byref(d))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (130, 70, 36), (300, 84, 64), (17, 81, 100), (1024, 512, 2560)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_layouts(dev, M, N, K, ta, tb):
    """mstts_gemm_bf16 = fp32-accumulated product of the bf16-ROUNDED operands (config 3), every layout incl. ragged edges."""
    A = _r(dev, *((K, M) if ta else (M, K)), seed=1)
    B = _r(dev, *((N, K) if tb else (K, N)), seed=2, scale=1.0 / np.sqrt(K))
    Cm = torch.zeros(M, N, device=dev)
    bias = _r(dev, N, seed=3)
    _gemm_call("mstts_gemm_bf16", A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, bias=bias, act=2)
    a = _bf(A).cpu().numpy(); b = _bf(B).cpu().numpy()
    ref = np.tanh((a.T if ta else a) @ (b.T if tb else b) + t2n(bias).astype(np.float64))
    assert rel_err(t2n(Cm), ref) < TOL
    if (M, N, K) == (256, 256, 128):          # ... and it really is a different product from the fp32 one
        C32 = torch.zeros(M, N, device=dev)
        _gemm_call("mstts_gemm_f32", A, B, C32, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, bias=bias, act=2)
        assert rel_err(t2n(C32), ref) > 1e-4


@pytest.mark.parametrize("M,N,K,ta,tb,sk", [(2048, 4096, 1030, 1, 0, 2), (6000, 2200, 520, 0, 0, 1), (3584, 4100, 333, 0, 1, 1), (2050, 3330, 96, 1, 1, 3)])
def test_gemm_bf16_big_tile(dev, M, N, K, ta, tb, sk):
    """The 256 x 256-tile bf16 kernel (round 5; taken when the output fills the chip with such tiles): every layout, ragged edges in M, N and K,
    unaligned (scalar-load) shapes, split-K atomics onto a pre-filled output, bias + activation; against the fp64 product of the bf16-ROUNDED
    operands, and equal to the 128 x 128 kernel's result to fp32 summation order."""
    A = _r(dev, *((K, M) if ta else (M, K)), seed=1)
    B = _r(dev, *((N, K) if tb else (K, N)), seed=2, scale=1.0 / np.sqrt(K))
    bias = _r(dev, N, seed=3)
    a = _bf(A).cpu().numpy(); b = _bf(B).cpu().numpy()
    prod = (a.T if ta else a) @ (b.T if tb else b)
    outs = {}
    for big in (1, 0):
        lib.load().mstts_gemm_bf16_big(big)
        if sk > 1:
            Cm = torch.ones(M, N, device=dev)
            _gemm_call("mstts_gemm_bf16", A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, split_k=sk, bias=bias)
            ref = 1.0 + prod + t2n(bias).astype(np.float64)
        else:
            Cm = torch.zeros(M, N, device=dev)
            _gemm_call("mstts_gemm_bf16", A, B, Cm, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb, bias=bias, act=2)
            ref = np.tanh(prod + t2n(bias).astype(np.float64))
        outs[big] = t2n(Cm).astype(np.float64)
        assert rel_err(outs[big], ref) < TOL, (big, rel_err(outs[big], ref))
    lib.load().mstts_gemm_bf16_big(1)
    assert rel_err(outs[1], outs[0]) < 2e-6


def test_gemm_bf16_big_tile_conv_window(dev):
    """... and its implicit-im2col window mode at the postnet's shape class (rows = batch x frames, 5 taps), forward and weight gradient."""
    B_, T, cin, cout, K = 40, 801, 64, 512, 5
    x = _r(dev, B_ * T, cin, seed=4); w = _r(dev, K * cin, cout, seed=5, scale=0.1)
    xb = _bf(x).cpu().reshape(B_, T, cin); wb = _bf(w).cpu().reshape(K, cin, cout)
    ref = torch.nn.functional.conv1d(xb.transpose(1, 2), wb.permute(2, 1, 0), padding=(K - 1) // 2).transpose(1, 2).reshape(B_ * T, cout).numpy()
    dy = _r(dev, B_ * T, cout, seed=6)
    xpad = torch.nn.functional.pad(xb, (0, 0, (K - 1) // 2, K - 1 - (K - 1) // 2))
    winm = xpad.unfold(1, K, 1).permute(0, 1, 3, 2).reshape(B_ * T, K * cin)
    refw = (winm.t() @ _bf(dy).cpu()).numpy()
    for big in (1, 0):
        lib.load().mstts_gemm_bf16_big(big)
        y = torch.zeros(B_ * T, cout, device=dev)
        _gemm_call("mstts_gemm_bf16", x, w, y, B_ * T, cout, K * cin, cin, cout, cout, win=(T, cin, (K - 1) // 2))
        assert rel_err(t2n(y), ref) < TOL, big
    lib.load().mstts_gemm_bf16_big(1)
    # weight gradient: M = K * cin = 320, N = 512 -> 4 tiles x split 64 = 256 workgroups of the big kernel
    dw = torch.zeros(K * cin, cout, device=dev)
    _gemm_call("mstts_gemm_bf16", x, dy, dw, K * cin, cout, B_ * T, cin, cout, cout, ta=1, win=(T, cin, (K - 1) // 2), split_k=64)
    assert rel_err(t2n(dw), refw) < TOL


def test_gemm_bf16_cuts_short_tile_lists_along_k(dev):
    """mstts_gemm_bf16 cuts a contraction whose 128 x 128 tile list is far from a round of the chip along K by itself (csrc/gemm_bf16.hip: 32
    tiles x K = 4 096 here): an uncut call WITHOUT `accumulate` must overwrite whatever the output held (the library clears it before the pieces
    add up), with `accumulate` it adds onto it, with a ragged output stride the rows' tails stay untouched, and under mstts_gemm_deterministic
    no cut is made: two runs are bit-equal and equal to the MSTTS_GEMM_BF16_AUTOCUT=0 form's summation order."""
    M, N, K, ldc = 512, 1000, 4096, 1024
    A = _r(dev, M, K, seed=1); B = _r(dev, K, N, seed=2, scale=1.0 / np.sqrt(K))
    bias = _r(dev, N, seed=3)
    ref = _bf(A).cpu().numpy().astype(np.float64) @ _bf(B).cpu().numpy().astype(np.float64) + t2n(bias).astype(np.float64)
    Cm = torch.full((M, ldc), 7.0, device=dev)                      # garbage in the output, a marker in the stride tail
    _gemm_call("mstts_gemm_bf16", A, B, Cm, M, N, K, K, N, ldc, bias=bias)
    assert rel_err(t2n(Cm)[:, :N], ref) < TOL
    assert float((Cm[:, N:] - 7.0).abs().max()) == 0.0
    Ca = torch.full((M, ldc), 2.0, device=dev)
    _gemm_call("mstts_gemm_bf16", A, B, Ca, M, N, K, K, N, ldc, bias=bias, accumulate=1)
    assert rel_err(t2n(Ca)[:, :N], 2.0 + ref) < TOL
    outs = []
    with lib.deterministic_gemm():
        for _ in range(2):
            Cd = torch.full((M, ldc), 7.0, device=dev)
            _gemm_call("mstts_gemm_bf16", A, B, Cd, M, N, K, K, N, ldc, bias=bias)
            outs.append(t2n(Cd)[:, :N].copy())
    assert np.array_equal(outs[0], outs[1])
    assert rel_err(outs[0], ref) < TOL


def test_gemm_bf16_honours_a_callers_cut(dev):
    """mstts_gemm_bf16 used to re-choose a caller's split_k for its own tiles (only an uncut call is the library's to cut now, like
    mstts_gemm_f32).  Observable: exactly TWO pieces onto a zeroed output are order-free (0 + p + q is the same float whichever atomic lands
    first), so ten runs are bit-equal - on the shape of the test above, which the library by itself cuts into more pieces than two."""
    M, N, K = 512, 1000, 4096
    A = _r(dev, M, K, seed=1); B = _r(dev, K, N, seed=2, scale=1.0 / np.sqrt(K))
    ref = _bf(A).cpu().numpy().astype(np.float64) @ _bf(B).cpu().numpy().astype(np.float64)
    first = None
    for _ in range(10):
        Cm = torch.zeros(M, N, device=dev)
        _gemm_call("mstts_gemm_bf16", A, B, Cm, M, N, K, K, N, N, split_k=2)
        if first is None:
            first = Cm.clone()
            assert rel_err(t2n(Cm), ref) < TOL
        assert torch.equal(Cm, first)


def test_gemm_bf16_conv_window_splitk(dev):
    """Implicit-im2col conv forward, weight gradient (transposed window, split-K atomics) and accumulate on the bf16 GEMM."""
    B_, T, cin, cout, K = 3, 37, 16, 24, 5
    x = _r(dev, B_ * T, cin, seed=4); w = _r(dev, K * cin, cout, seed=5, scale=0.2)
    y = torch.zeros(B_ * T, cout, device=dev)
    _gemm_call("mstts_gemm_bf16", x, w, y, B_ * T, cout, K * cin, cin, cout, cout, win=(T, cin, (K - 1) // 2))
    xb = _bf(x).cpu().reshape(B_, T, cin); wb = _bf(w).cpu().reshape(K, cin, cout)
    ref = torch.nn.functional.conv1d(xb.transpose(1, 2), wb.permute(2, 1, 0), padding=(K - 1) // 2).transpose(1, 2).reshape(B_ * T, cout)
    assert rel_err(t2n(y), ref.numpy()) < TOL
    dy = _r(dev, B_ * T, cout, seed=6)
    dw = torch.ones(K * cin, cout, device=dev)            # accumulate onto ones through split-K atomics
    _gemm_call("mstts_gemm_bf16", x, dy, dw, K * cin, cout, B_ * T, cin, cout, cout, ta=1, win=(T, cin, (K - 1) // 2), split_k=3)
    xpad = torch.nn.functional.pad(xb, (0, 0, (K - 1) // 2, K - 1 - (K - 1) // 2))
    winm = xpad.unfold(1, K, 1).permute(0, 1, 3, 2).reshape(B_ * T, K * cin)
    refw = 1.0 + winm.t() @ _bf(dy).cpu()
    assert rel_err(t2n(dw), refw.numpy()) < TOL


@pytest.mark.parametrize("M,N,K", [(32, 4096, 1792), (32, 4096, 2048), (32, 128, 1024), (5, 256, 192), (40, 132, 64)])
def test_skinny_fwd(dev, M, N, K):
    L = lib.load()
    ks = L.mstts_skinny_fwd_splits(N, K)
    assert ks >= 1
    X = _r(dev, M, K + 4, seed=1)[:, :K]          # row stride K+4: exercises ldx != K
    W = _r(dev, K, N, seed=2, scale=0.1)
    P = torch.zeros(ks, M, N, device=dev)
    lib.call("mstts_skinny_fwd", lib.ptr(X), K + 4, lib.ptr(W), N, lib.ptr(P), 0, M, N, K, ks)
    ref = t2n(X).astype(np.float64) @ t2n(W).astype(np.float64)
    assert rel_err(t2n(P).astype(np.float64).sum(0), ref) < TOL


@pytest.mark.parametrize("M,R,N", [(32, 1792, 4096), (32, 2048, 4096), (32, 1024, 128), (7, 192, 256), (33, 40, 64)])
def test_skinny_bwd(dev, M, R, N):
    L = lib.load()
    ns = L.mstts_skinny_bwd_splits(R, N)
    assert ns >= 1
    dG = _r(dev, M, N, seed=3)
    W = _r(dev, R, N, seed=4, scale=0.1)
    P = torch.zeros(ns, M, R, device=dev)
    lib.call("mstts_skinny_bwd", lib.ptr(dG), N, lib.ptr(W), N, lib.ptr(P), 0, M, R, N, ns)
    ref = t2n(dG).astype(np.float64) @ t2n(W).astype(np.float64).T
    assert rel_err(t2n(P).astype(np.float64).sum(0), ref) < TOL
    if R % 32 == 0:          # the same product against the packed kernel: identical arithmetic, bit-identical slabs
        Wp = torch.zeros(R * N, device=dev)
        lib.call("mstts_pack_skinny_bwd", lib.ptr(W), N, lib.ptr(Wp), R, N, ns)
        P2 = torch.zeros(ns, M, R, device=dev)
        lib.call("mstts_skinny_bwd_packed", lib.ptr(dG), N, lib.ptr(Wp), lib.ptr(P2), 0, M, R, N, ns)
        assert torch.equal(P, P2)


@pytest.mark.parametrize("B,H,K,mode", [(32, 1024, 1792, "xw"), (32, 1024, 2048, "bias"), (5, 64, 192, "xw"), (17, 8, 64, "bias"), (40, 16, 128, "none"),
                                         (16, 1024, 2048, "bias")])
def test_cell_fwd_fused(dev, B, H, K, mode):
    """Fused cell step (gates product + zoneout-LSTM update in one launch, packed kernel) vs the cell in fp64
    (ZoneoutLSTMCell.py:228-271): output m, zoned state, saved activations / raw cell state."""
    L = lib.load()
    assert L.mstts_cell_fwd_supported(H, K) == 1 and L.mstts_cell_fwd_supported(H, K + 16) == 0
    ldx, hld, old = K + 8, H + 4, H + 12
    X = _r(dev, B, ldx, seed=1)
    W = _r(dev, K, 4 * H, seed=2, scale=1.0 / np.sqrt(K))
    Wp = torch.zeros(K * 4 * H, device=dev)
    lib.call("mstts_pack_cell_fwd", lib.ptr(W), 4 * H, lib.ptr(Wp), K, H)
    assert sorted(t2n(Wp).tolist()) == sorted(t2n(W).reshape(-1).tolist())          # a permutation of the kernel
    xw = _r(dev, B, 4 * H, seed=3) if mode == "xw" else None
    bias = _r(dev, 4 * H, seed=4, scale=0.3) if mode == "bias" else None
    cp, hp = _r(dev, B, H, seed=5), _r(dev, B, hld, seed=6)
    g = np.random.default_rng(7)
    zc = torch.tensor(g.integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    zh = torch.tensor(g.integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    out, cn, hn = torch.zeros(B, old, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, hld, device=dev)
    acts, craw = torch.zeros(B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
    Xp = torch.full((int(L.mstts_cell_act_floats(B, K)),), float("nan"), device=dev)
    lib.call("mstts_pack_cell_act", lib.ptr(X), ldx, lib.ptr(Xp), B, K)
    assert sorted(t2n(Xp)[t2n(Xp) != 0].tolist()) == sorted(t2n(X)[:, :K].reshape(-1).tolist())       # a permutation of the block (+ zero rows)
    K2, c0 = 2 * K if 2 * K <= 2048 else K, 64 if H + 64 <= K else 0
    outp = torch.zeros(int(L.mstts_cell_act_floats(B, K2)), device=dev)
    hnp = torch.zeros(int(L.mstts_cell_act_floats(B, K)), device=dev)
    d = lib.CellFwd()
    d.B, d.H, d.K, d.Xp, d.Wp = B, H, K, lib.ptr(Xp), lib.ptr(Wp)
    if H <= K:               # packed copies of m (into a block of width K2 at column 0) and h' (width K at column c0)
        d.out_p.base, d.out_p.K, d.out_p.col0 = lib.ptr(outp), K2, 0
        if c0 + H <= K:
            d.h_next_p.base, d.h_next_p.K, d.h_next_p.col0 = lib.ptr(hnp), K, c0
    d.xw, d.xw_ld, d.bias = lib.ptr(xw), 4 * H, lib.ptr(bias)
    d.c_prev, d.h_prev, d.h_prev_ld, d.zc, d.zh, d.zoneout = lib.ptr(cp), lib.ptr(hp), hld, lib.ptr(zc), lib.ptr(zh), 0.1
    d.out, d.out_ld, d.c_next, d.h_next, d.h_next_ld, d.acts, d.c_raw = lib.ptr(out), old, lib.ptr(cn), lib.ptr(hn), hld, lib.ptr(acts), lib.ptr(craw)
    lib.call("mstts_cell_fwd", C.byref(d))
    gates = t2n(X)[:, :K].astype(np.float64) @ t2n(W).astype(np.float64)
    if xw is not None:
        gates = gates + t2n(xw)
    if bias is not None:
        gates = gates + t2n(bias)
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    i, j, f, o = np.split(gates, 4, axis=1)
    c = sg(f + 1.0) * t2n(cp) + sg(i) * np.tanh(j)
    m = sg(o) * np.tanh(c)
    c2 = 0.9 * t2n(zc) * (c - t2n(cp)) + t2n(cp)
    h2 = 0.9 * t2n(zh) * (m - t2n(hp)[:, :H]) + t2n(hp)[:, :H]
    assert rel_err(t2n(out)[:, :H], m) < TOL and rel_err(t2n(cn), c2) < TOL and rel_err(t2n(hn)[:, :H], h2) < TOL
    assert float(out[:, H:].abs().max()) == 0.0 and float(hn[:, H:].abs().max()) == 0.0                # nothing written past the H columns
    assert rel_err(t2n(craw), c) < TOL and rel_err(t2n(acts), np.concatenate([sg(i), np.tanh(j), sg(f + 1.0), sg(o)], 1)) < TOL
    if H <= K:               # the packed copies hold exactly what the row-major outputs hold
        ref = torch.zeros(B, K2, device=dev); ref[:, :H] = out[:, :H]
        chk = torch.zeros_like(outp)
        lib.call("mstts_pack_cell_act", lib.ptr(ref), K2, lib.ptr(chk), B, K2)
        assert torch.equal(chk, outp)
        if c0 + H <= K:
            ref = torch.zeros(B, K, device=dev); ref[:, c0:c0 + H] = hn[:, :H]
            chk = torch.zeros_like(hnp)
            lib.call("mstts_pack_cell_act", lib.ptr(ref), K, lib.ptr(chk), B, K)
            assert torch.equal(chk, hnp)




@pytest.mark.parametrize("M,N,K", [(32, 4096, 1792), (32, 4096, 2048), (16, 128, 1024), (5, 256, 192), (32, 64, 64), (40, 128, 512)])
def test_skinny_bf16_fwd(dev, M, N, K):
    """bf16 weight-streaming product: bf(X) . bf(W) with fp32 accumulation (packed kernel, K-split partial slabs)."""
    L = lib.load()
    ks = L.mstts_skinny_bf16_fwd_splits(N, K)
    assert ks >= 1
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.zeros(M, K + 8); X[:, :K] = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) * 0.05
    Xd, Wd = X.to(dev), W.to(dev)
    Wp = torch.zeros(K * N, dtype=torch.int16, device=dev)
    lib.call("mstts_pack_bf16_fwd", lib.ptr(Wd), N, lib.ptr(Wp), K, N, ks)
    P = torch.zeros(ks, M, N, device=dev)
    lib.call("mstts_skinny_fwd_bf16", lib.ptr(Xd), K + 8, lib.ptr(Wp), lib.ptr(P), 0, M, N, K, ks)
    ref = _bf(X[:, :K]) @ _bf(W)
    got = P.sum(0).double().cpu()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    assert sorted(Wp.cpu().numpy().astype(np.uint16).tolist()) == sorted(W.to(torch.bfloat16).view(torch.int16).flatten().numpy().astype(np.uint16).tolist())


@pytest.mark.parametrize("M,R,N", [(32, 1792, 4096), (32, 2048, 4096), (16, 1024, 128), (5, 96, 256), (32, 32, 64), (40, 64, 1024)])
def test_skinny_bf16_bwd(dev, M, R, N):
    """bf16 data-gradient product: bf(dG) . bf(W)^T with fp32 accumulation."""
    L = lib.load()
    ns = L.mstts_skinny_bf16_bwd_splits(R, N)
    assert ns >= 1
    g = torch.Generator().manual_seed(M + N + R)
    dG = torch.randn(M, N, generator=g); W = torch.randn(R, N, generator=g) * 0.05
    dGd, Wd = dG.to(dev), W.to(dev)
    Wq = torch.zeros(R * N, dtype=torch.int16, device=dev)
    lib.call("mstts_pack_bf16_bwd", lib.ptr(Wd), N, lib.ptr(Wq), R, N, ns)
    P = torch.zeros(ns, M, R, device=dev)
    lib.call("mstts_skinny_bwd_bf16", lib.ptr(dGd), N, lib.ptr(Wq), lib.ptr(P), 0, M, R, N, ns)
    ref = _bf(dG) @ _bf(W).t()
    got = P.sum(0).double().cpu()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("B,T,H", [(5, 9, 64), (32, 12, 256)])
def test_lstm_seq_fused_and_pair_forms(dev, B, T, H):
    """dynamic_rnn drivers: the fused-step form (one mstts_cell_fwd launch per step, packed recurrent kernel / packed h) and the
    two-directions-per-launch pair forms give what the plain product + pointwise launches give - ragged lengths, reversed direction,
    outputs / histories / BPTT saves forward; dgates / d_h chains backward."""
    L = lib.load()
    assert L.mstts_cell_fwd_supported(H, H) == 1
    g = np.random.default_rng(3)
    lengths = torch.tensor(np.concatenate([[T], g.integers(1, T + 1, B - 1)]).astype(np.int32), device=dev)
    BH = B * H

    def make(direction, fused):
        xw = _r(dev, B, T, 4 * H, seed=10 + direction)
        wh = _r(dev, H, 4 * H, seed=20 + direction, scale=1.0 / np.sqrt(H))
        zc = torch.tensor(np.random.default_rng(30 + direction).integers(0, 2, (T, B, H)).astype(np.uint8), device=dev)
        zh = torch.tensor(np.random.default_rng(40 + direction).integers(0, 2, (T, B, H)).astype(np.uint8), device=dev)
        t = dict(xw=xw, wh=wh, zc=zc, zh=zh, out=torch.zeros(B, T, 2 * H, device=dev), c=torch.zeros(T + 1, B, H, device=dev),
                 h=torch.zeros(T + 1, B, H, device=dev), acts=torch.zeros(T, B, 4 * H, device=dev), craw=torch.zeros(T, B, H, device=dev),
                 ws=torch.zeros(int(L.mstts_lstm_seq_ws_floats(B, H, 0)), device=dev))
        q = lib.LstmSeqFwd()
        q.B, q.T, q.H = B, T, H
        q.xw, q.wh, q.wh_ld, q.lengths, q.reverse, q.zoneout = lib.ptr(xw), lib.ptr(wh), 4 * H, lib.ptr(lengths), direction, 0.1
        q.zc, q.zh = lib.ptr(zc), lib.ptr(zh)
        q.out, q.out_sb, q.out_st = lib.ptr(t["out"], direction * H), T * 2 * H, 2 * H
        q.c_hist, q.h_hist, q.acts, q.c_raw, q.gates_ws = lib.ptr(t["c"]), lib.ptr(t["h"]), lib.ptr(t["acts"]), lib.ptr(t["craw"]), lib.ptr(t["ws"])
        if fused:
            t["whp"] = torch.zeros(H * 4 * H, device=dev)
            lib.call("mstts_pack_cell_fwd", lib.ptr(wh), 4 * H, lib.ptr(t["whp"]), H, H)
            t["hp"] = torch.zeros(2 * int(L.mstts_cell_act_floats(B, H)), device=dev)
            q.wh_p, q.h_p = lib.ptr(t["whp"]), lib.ptr(t["hp"])
        return q, t

    ref = [make(0, False), make(1, False)]
    for q, _ in ref:
        lib.call("mstts_lstm_seq_fwd", C.byref(q))
    one = [make(0, True), make(1, True)]
    for q, _ in one:
        lib.call("mstts_lstm_seq_fwd", C.byref(q))                         # fused steps, one direction at a time
    pair = [make(0, True), make(1, True)]
    lib.call("mstts_lstm_seq_fwd_pair", C.byref(pair[0][0]), C.byref(pair[1][0]))
    torch.cuda.synchronize()
    for (_, a), (_, b), (_, c_) in zip(ref, one, pair):
        for k in ("out", "c", "h", "acts", "craw"):
            assert rel_err(t2n(b[k]), t2n(a[k])) < TOL and torch.equal(b[k], c_[k]), k
    # backward: pair driver vs two single calls on the same saved forward
    def make_bwd(direction, fwd):
        t = dict(dout=_r(dev, B, T, 2 * H, seed=50 + direction), dgs=torch.zeros(T, B, 4 * H, device=dev), dgp=torch.zeros(B, T, 4 * H, device=dev),
                 ws=torch.zeros(int(L.mstts_lstm_seq_ws_floats(B, H, 1)), device=dev))
        q = lib.LstmSeqBwd()
        q.B, q.T, q.H = B, T, H
        q.wh, q.wh_ld, q.lengths, q.reverse, q.zoneout = lib.ptr(fwd["wh"]), 4 * H, lib.ptr(lengths), direction, 0.1
        q.zc, q.zh = lib.ptr(fwd["zc"]), lib.ptr(fwd["zh"])
        q.d_out, q.dout_sb, q.dout_st = lib.ptr(t["dout"], direction * H), T * 2 * H, 2 * H
        q.c_hist, q.acts, q.c_raw = lib.ptr(fwd["c"]), lib.ptr(fwd["acts"]), lib.ptr(fwd["craw"])
        q.dgates_step, q.dgates_pos, q.ws = lib.ptr(t["dgs"]), lib.ptr(t["dgp"]), lib.ptr(t["ws"])
        return q, t
    single = [make_bwd(0, ref[0][1]), make_bwd(1, ref[1][1])]
    for q, _ in single:
        lib.call("mstts_lstm_seq_bwd", C.byref(q))
    both = [make_bwd(0, ref[0][1]), make_bwd(1, ref[1][1])]
    lib.call("mstts_lstm_seq_bwd_pair", C.byref(both[0][0]), C.byref(both[1][0]))
    torch.cuda.synchronize()
    for (_, a), (_, b) in zip(single, both):
        assert float(a["dgs"].abs().max()) > 0 and torch.equal(a["dgs"], b["dgs"]) and torch.equal(a["dgp"], b["dgp"])


@pytest.mark.parametrize("B,H,K", [(32, 1024, 1792), (32, 1024, 2048), (9, 32, 128), (20, 64, 384)])
def test_cell_fwd_fused_bf16(dev, B, H, K):
    """bf16 form of the fused cell step (config 3): bf16 copies of kernel and activation block, v_mfma_f32_16x16x32_bf16 with fp32
    accumulation - equals the cell evaluated in fp64 on the bf16-ROUNDED operands; the packed bf16 copies it writes for the next
    cells hold exactly the rounded outputs."""
    L = lib.load()
    assert L.mstts_cell_fwd_bf16_supported(H, K) == 1 and L.mstts_cell_fwd_bf16_supported(H, K + 64) == 0
    X = _r(dev, B, K + 8, seed=1)
    W = _r(dev, K, 4 * H, seed=2, scale=1.0 / np.sqrt(K))
    Wp = torch.zeros(K * 4 * H, dtype=torch.int16, device=dev)
    Xp = torch.zeros(int(L.mstts_cell_act_floats(B, K)), dtype=torch.int16, device=dev)
    lib.call("mstts_pack_cell_fwd_bf16", lib.ptr(W), 4 * H, lib.ptr(Wp), K, H)
    lib.call("mstts_pack_cell_act_bf16", lib.ptr(X), K + 8, lib.ptr(Xp), B, K)
    bias = _r(dev, 4 * H, seed=4, scale=0.3)
    cp, hp = _r(dev, B, H, seed=5), _r(dev, B, H, seed=6)
    g = np.random.default_rng(7)
    zc = torch.tensor(g.integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    zh = torch.tensor(g.integers(0, 2, (B, H)).astype(np.uint8), device=dev)
    out, cn, hn = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
    acts, craw = torch.zeros(B, 4 * H, device=dev), torch.zeros(B, H, device=dev)
    K2 = 2 * K if 2 * K <= 2048 else K
    outp = torch.zeros(int(L.mstts_cell_act_floats(B, K2)), dtype=torch.int16, device=dev)
    d = lib.CellFwd()
    d.B, d.H, d.K, d.Xp, d.Wp, d.bias, d.bf16 = B, H, K, lib.ptr(Xp), lib.ptr(Wp), lib.ptr(bias), 1
    d.c_prev, d.h_prev, d.h_prev_ld, d.zc, d.zh, d.zoneout = lib.ptr(cp), lib.ptr(hp), H, lib.ptr(zc), lib.ptr(zh), 0.1
    d.out, d.out_ld, d.c_next, d.h_next, d.h_next_ld, d.acts, d.c_raw = lib.ptr(out), H, lib.ptr(cn), lib.ptr(hn), H, lib.ptr(acts), lib.ptr(craw)
    if H <= K2:
        d.out_p.base, d.out_p.K, d.out_p.col0, d.out_p.bf16 = lib.ptr(outp), K2, 0, 1
    lib.call("mstts_cell_fwd", C.byref(d))
    gates = _bf(X[:, :K]).cpu().numpy() @ _bf(W).cpu().numpy() + t2n(bias)
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    i, j, f, o = np.split(gates, 4, axis=1)
    c = sg(f + 1.0) * t2n(cp) + sg(i) * np.tanh(j)
    m = sg(o) * np.tanh(c)
    assert rel_err(t2n(out), m) < TOL and rel_err(t2n(cn), 0.9 * t2n(zc) * (c - t2n(cp)) + t2n(cp)) < TOL
    assert rel_err(t2n(hn), 0.9 * t2n(zh) * (m - t2n(hp)) + t2n(hp)) < TOL and rel_err(t2n(craw), c) < TOL
    if H <= K2:
        ref = torch.zeros(B, K2, device=dev); ref[:, :H] = out
        chk = torch.zeros_like(outp)
        lib.call("mstts_pack_cell_act_bf16", lib.ptr(ref), K2, lib.ptr(chk), B, K2)
        assert torch.equal(chk, outp)


def test_bf16_exchange_kernels(dev):
    """mstts_f32_to_bf16 (round to nearest even) / mstts_bf16_chunks_sum (fp32 accumulation in chunk order, one rounding) /
    mstts_bf16_to_f32: the three kernels of the config-3 gradient exchange, against torch's bf16 conversions bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(4, 1003, generator=g) * torch.logspace(-6, 3, 1003)).to(dev)
    x[0, :4] = torch.tensor([1.0 + 2 ** -9, 1.0 + 3 * 2 ** -9, -0.0, 65504.0], device=dev)      # ties to even, signed zero, large
    y = torch.zeros(4, 1003, dtype=torch.bfloat16, device=dev)
    lib.call("mstts_f32_to_bf16", lib.ptr(x), lib.ptr(y), x.numel())
    assert torch.equal(y, x.to(torch.bfloat16))
    s = torch.zeros(1003, dtype=torch.bfloat16, device=dev)
    lib.call("mstts_bf16_chunks_sum", lib.ptr(y), 4, 1003, 1003, lib.ptr(s))
    want = ((y[0].float() + y[1].float()) + y[2].float() + y[3].float()).to(torch.bfloat16)
    assert torch.equal(s, want)
    z = torch.zeros(1003, device=dev)
    lib.call("mstts_bf16_to_f32", lib.ptr(s), lib.ptr(z), 1003)
    assert torch.equal(z, s.float())
