"""The WaveGlow dilated convolution [rows, 3 x 512] x [1536, 1024] as ONE reduction piece accumulated onto the conditioning buffer (128 x 128-tile
kernel) against TWO pieces onto a zeroed buffer (the 256 x 256-tile kernel where 2 x its tile count reaches 160 workgroups), zero fill and the wider
gate included: microseconds per layer for batches of 40-frame chunks.   usage: python tools/wg_conv_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
from multi_speaker_tts_amd.lib import call, gemm, ptr

dev = torch.device("cuda:0")
lib.load()
ch, k, Lg = 512, 3, 1376
w = torch.randn(k * ch, 2 * ch, device=dev) * 0.02


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


with lib.deterministic_gemm():
    for N in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
        rows = N * Lg
        x = torch.randn(rows, ch, device=dev)
        cond = torch.randn(rows, 8 * 2 * ch, device=dev)
        conv = torch.empty(rows, 2 * ch, device=dev)
        z = torch.empty(rows, ch, device=dev)
        win = (Lg, ch, 1, 4)

        def one():
            gemm(x, w, cond, rows, 2 * ch, k * ch, ch, 2 * ch, 8 * 2 * ch, accumulate=True, win=win, c_off=2 * 2 * ch)
            call("mstts_wg_gate", ptr(cond, 2 * 2 * ch), 8 * 2 * ch, ptr(z), rows, ch)

        def two():
            conv.zero_()
            gemm(x, w, conv, rows, 2 * ch, k * ch, ch, 2 * ch, 2 * ch, split_k=2, win=win)
            call("mstts_wg_gate_add", ptr(cond, 2 * 2 * ch), 8 * 2 * ch, ptr(conv), ptr(z), rows, ch)

        a, b = timed(one), timed(two)
        t128 = -(-rows // 128) * 8
        w256 = -(-rows // 256) * 4 * 2
        print("batch %2d rows %6d: one piece %7.1f us (%4d tiles of 128^2 = %.2f rounds) | two pieces %7.1f us (%4d workgroups of 256^2 = %.2f rounds)  %s"
              % (N, rows, a, t128, t128 / 256, b, w256, w256 / 256, "TWO" if b < a else "one"))
