import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims
dev = torch.device("cuda:0")
d = Dims()
eng = TrainEngine(d, device=dev, seed=1)
batch = bench.synthetic_batch(d, 32, 128, 800, 1, 0, dev)
t0 = time.time()
bad = 0
for i in range(40):
    w = eng.train_step(batch)
    s = eng.scalars(w)
    cnt = int(w.energy_ws.view(torch.int64)[32 * 128].item()) if hasattr(w.energy_ws, "view") else -1
    ok = np.isfinite(s["Loss"]) and bool(torch.isfinite(eng.params.grad).all())
    bad += (not ok) or cnt != 0
    if i % 10 == 0:
        print(i, s["Loss"], "timeouts", cnt, "finite", ok, flush=True)
print("steps 40, bad", bad, "elapsed %.1f s" % (time.time() - t0))
