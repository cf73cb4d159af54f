"""Every distinct large contraction of one config-2 train step timed on its own (HIP events, 10 repeats): shape, layout, split-K ->
microseconds and TFLOP/s, fp32 (mstts_gemm_f32) and bf16-operand (mstts_gemm_bf16) kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.engine import _split_k

dev = torch.device("cuda:0")
B, Te, S = 32, 128, 801
SB, BT = S * B, B * Te
shapes = [  # name, M, N, K, trans_a, trans_b, win(T, C, pad) or None, split_k, count per step
    ("postnet conv fwd 512->512", SB, 512, 2560, 0, 0, (S, 512, 2), 1, 3),
    ("postnet conv dgrad 512->512", SB, 512, 2560, 0, 0, (S, 512, 2), 1, 3),
    ("postnet conv wgrad 512->512", 2560, 512, SB, 1, 0, (S, 512, 2), max(2, _split_k(2560, 512, SB)), 3),
    ("  (same shape, plain A, no window)", SB, 512, 2560, 0, 0, None, 1, 0),
    ("postnet conv fwd 80->512", SB, 512, 400, 0, 0, (S, 80, 2), 1, 1),
    ("postnet conv fwd 512->80", SB, 80, 2560, 0, 0, (S, 512, 2), 1, 1),
    ("postnet conv wgrad 512->80", 2560, 80, SB, 1, 0, (S, 512, 2), max(2, _split_k(2560, 80, SB)), 1),
    ("encoder conv fwd", BT, 512, 2560, 0, 0, (Te, 512, 2), 1, 3),
    ("encoder conv wgrad", 2560, 512, BT, 1, 0, (Te, 512, 2), max(2, _split_k(2560, 512, BT)), 3),
    ("xw0 = prenet . W0[:P]", SB, 4096, 256, 0, 0, None, 1, 1),
    ("d_pre = dg0 . W0[:P]^T", SB, 256, 4096, 0, 1, None, 1, 1),
    ("dW1 = in1^T . dg1", 2048, 4096, SB, 1, 0, None, _split_k(2048, 4096, SB), 1),
    ("dw0f = in0^T . dg0", 1792, 4096, SB, 1, 0, None, _split_k(1792, 4096, SB), 1),
    ("dW0[:P] = pre^T . dg0", 256, 4096, SB, 1, 0, None, max(2, _split_k(256, 4096, SB)), 1),
    ("projection fwd", SB, 84, 1792, 0, 0, None, 1, 1),
    ("d_pj = d_proj . Wp^T", SB, 1792, 84, 0, 1, None, 1, 1),
    ("dWp = pj^T . d_proj", 1792, 84, SB, 1, 0, None, max(2, _split_k(1792, 84, SB)), 1),
    ("dWq = m1^T . dq", 1024, 128, SB, 1, 0, None, max(2, _split_k(1024, 128, SB)), 1),
    ("prenet_1 fwd", SB, 256, 256, 0, 0, None, 1, 1),
]
if len(sys.argv) > 1:                         # python tools/gemm_shapes_bench.py 0 : body + tail scheduling off (A/B)
    lib.call("mstts_gemm_tail_split", int(sys.argv[1]))
tot = {"f32": 0.0, "bf16": 0.0}
print("%-34s %7s %6s %7s sk | %9s %8s | %9s %8s" % ("contraction", "M", "N", "K", "f32 us", "TFLOP/s", "bf16 us", "TFLOP/s"))
for name, M, N, K, ta, tb, win, sk, cnt in shapes:
    if win:
        A = torch.randn(M if not ta else K, win[1], device=dev)          # X[rows, C]; the kernel forms the windows
        lda = win[1]
    else:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        lda = A.shape[1]
    Bm = torch.randn((N, K) if tb else (K, N), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    res = []
    for bf in (False, True):
        for _ in range(2):
            lib.gemm(A, Bm, Cm, M, N, K, lda, Bm.shape[1], N, trans_a=bool(ta), trans_b=bool(tb), win=win, split_k=sk, bf16=bf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.gemm(A, Bm, Cm, M, N, K, lda, Bm.shape[1], N, trans_a=bool(ta), trans_b=bool(tb), win=win, split_k=sk, bf16=bf)
        e1.record()
        torch.cuda.synchronize()
        us = 100.0 * e0.elapsed_time(e1)
        res.append((us, 2.0 * M * N * K / us / 1e6))
        tot["bf16" if bf else "f32"] += us * cnt
    print("%-34s %7d %6d %7d %2d | %9.1f %8.1f | %9.1f %8.1f   x%d" % (name, M, N, K, sk, res[0][0], res[0][1], res[1][0], res[1][1], cnt))
print("sum per step: f32 %.2f ms, bf16 %.2f ms" % (tot["f32"] / 1e3, tot["bf16"] / 1e3))
