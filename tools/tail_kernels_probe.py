"""The step's scalar streaming kernels alone on the chip (HIP events, 50 repeats): Adam over the step's 30.3 M variables, the regulariser's sum, a torch copy of the same bytes.
usage: python tools/tail_kernels_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_speaker_tts_amd import lib
dev = torch.device("cuda:0")
n = 30279680          # the step's variables (4 per work-item of the gradient slab's fill: 7 569 920 work-items)
p, g, m, v = [torch.randn(n, device=dev) * 0.01 for _ in range(4)]
v = v.abs()
wd = (torch.rand(n, device=dev) > 0.5).to(torch.uint8)
out = torch.zeros(4, device=dev)
def timed(f, reps=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
t = timed(lambda: lib.call("mstts_adam_tf", lib.ptr(p), lib.ptr(g), lib.ptr(m), lib.ptr(v), lib.ptr(wd), 1e-6, 1.0, 1e-4, 0.9, 0.999, 1e-6, n))
print("adam_tf            %7.1f us  %6.2f TB/s (7 x 4 B + 1 B per element)" % (t, n * 29 / t * 1e-6))
t = timed(lambda: lib.call("mstts_l2_loss_acc", lib.ptr(p), lib.ptr(wd), n, lib.ptr(out)))
print("l2_loss_acc        %7.1f us  %6.2f TB/s (5 B per element)" % (t, n * 5 / t * 1e-6))
q = torch.empty_like(p)
t = timed(lambda: q.copy_(p))
print("torch copy         %7.1f us  %6.2f TB/s (8 B per element)" % (t, n * 8 / t * 1e-6))
t = timed(lambda: torch.add(p, g, out=q))
print("torch add          %7.1f us  %6.2f TB/s (12 B per element)" % (t, n * 12 / t * 1e-6))
