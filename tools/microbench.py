"""Launch-floor / kernel-floor microbenchmarks on one MI355X (development aid)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib

dev = torch.device("cuda:0")
L = lib.load()


def timeit(fn, n, graph=False):
    fn(); torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    fn()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


x = torch.zeros(1 << 20, device=dev)
small = lambda: lib.call("mstts_fill", lib.ptr(x), 1.0, 1024)
print("tiny fill kernel, eager     : %.2f us/launch" % timeit(small, 2000))
print("tiny fill kernel, hipGraph  : %.2f us/launch" % timeit(small, 2000, graph=True))
a = torch.randn(1 << 16, device=dev); b = torch.randn(1 << 16, device=dev); y = torch.zeros(1 << 16, device=dev)
add = lambda: lib.call("mstts_add", lib.ptr(a), lib.ptr(b), lib.ptr(y), 1 << 16)
print("add 64K (3 arrays), eager   : %.2f us" % timeit(add, 2000))
print("add 64K, hipGraph           : %.2f us" % timeit(add, 2000, graph=True))
# dependent chain add: y = y + b
addc = lambda: lib.call("mstts_add", lib.ptr(y), lib.ptr(b), lib.ptr(y), 1 << 16)
print("dependent add chain, graph  : %.2f us" % timeit(addc, 2000, graph=True))

M, N, K = 32, 4096, 1792
X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.02
ks = L.mstts_skinny_fwd_splits(N, K)
P = torch.zeros(ks, M, N, device=dev)
sk = lambda: lib.call("mstts_skinny_fwd", lib.ptr(X), K, lib.ptr(W), N, lib.ptr(P), 0, M, N, K, ks)
print("skinny_fwd 32x1792x4096 (ks=%d), eager : %.2f us" % (ks, timeit(sk, 1000)))
print("skinny_fwd, hipGraph                  : %.2f us" % timeit(sk, 1000, graph=True))
# several weight copies to defeat cache residency
Ws = [torch.randn(K, N, device=dev) * 0.02 for _ in range(8)]
def sk_rot(i=[0]):
    i[0] = (i[0] + 1) % 8
    lib.call("mstts_skinny_fwd", lib.ptr(X), K, lib.ptr(Ws[i[0]]), N, lib.ptr(P), 0, M, N, K, ks)
print("skinny_fwd rotating 8 weight sets (235 MB), graph : %.2f us" % timeit(sk_rot, 1000, graph=True))
R = 1792
dG = torch.randn(M, N, device=dev); Wb = torch.randn(R, N, device=dev) * 0.02
ns = L.mstts_skinny_bwd_splits(R, N)
Pb = torch.zeros(ns, M, R, device=dev)
skb = lambda: lib.call("mstts_skinny_bwd", lib.ptr(dG), N, lib.ptr(Wb), N, lib.ptr(Pb), 0, M, R, N, ns)
print("skinny_bwd 32x4096x1792 (ns=%d), graph : %.2f us" % (ns, timeit(skb, 1000, graph=True)))
# generic 128-tile GEMM throughput
for (m, n, k) in [(25632, 4096, 256), (25632, 512, 2560), (4096, 4096, 4096)]:
    A = torch.randn(m, k, device=dev); Bm = torch.randn(k, n, device=dev); Cm = torch.zeros(m, n, device=dev)
    f = lambda: lib.gemm(A, Bm, Cm, m, n, k, k, n, n)
    us = timeit(f, 20)
    print("gemm NN %dx%dx%d: %.1f us  %.1f TFLOP/s" % (m, n, k, us, 2.0 * m * n * k / us / 1e6))
m, n, k = 2048, 4096, 25632
A = torch.randn(k, m, device=dev); Bm = torch.randn(k, n, device=dev); Cm = torch.zeros(m, n, device=dev)
f = lambda: lib.gemm(A, Bm, Cm, m, n, k, m, n, n, trans_a=True)
us = timeit(f, 10)
print("gemm TN %dx%dx%d: %.1f us  %.1f TFLOP/s" % (m, n, k, us, 2.0 * m * n * k / us / 1e6))
