"""Does a hipGraph replay of the whole train step beat eager launches?  (timing probe only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims

dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dims = Dims()
eng = TrainEngine(dims, device=dev)
batch = bench.synthetic_batch(dims, 32, 128, L, 1234, 0, dev)
for _ in range(2):
    eng.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.train_step(batch)
torch.cuda.synchronize()
print("eager  : %.1f ms/step" % ((time.perf_counter() - t0) / 3 * 1e3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        eng.train_step(batch)
torch.cuda.synchronize()
print("capture+instantiate: %.2f s" % (time.perf_counter() - t0))
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("graph  : %.1f ms/step" % ((time.perf_counter() - t0) / 3 * 1e3))
