"""Per-workgroup picture of the persistent forward launch's stage stamps (one extra step with the stamping instantiation): for every stage the
mean / min / max over the 256 workgroups, and the means by contraction slice gi = g & 7 (= XCD) and by column slice / attention row gj = g >> 3.
A stage whose time differs systematically between workgroups shows who the others wait for."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd.params import Dims

dev = torch.device("cuda:0")
dims = Dims()
C3 = "--config3" in sys.argv            # the bf16 instantiations of the two loops
eng = TrainEngine(dims, device=dev, seed=1234, recurrent_dtype="bf16" if C3 else "f32", gemm_dtype="bf16" if C3 else "f32")
batch = bench.synthetic_batch(dims, 32, 128, 800, 1234, 0, dev)
w = eng.plan(32, 128, 800)
for _ in range(2):
    eng.forward(batch, w); eng.loss_and_backward(w); eng.adam_step()
names = ["loop top+prenet", "wait ctx", "ctx product", "wait partials0", "update0", "wait m0", "m0 product", "shadow h0 1", "wait partials1", "update1",
         "shadow h0 2", "wait m1", "query+energies", "shadow h1", "wait energies", "softmax+ctx"]
order = [0, 1, 2, 13, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15]
for which in ("forward", "bptt"):
    st = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    if which == "forward":
        eng.persist_stamps = st
        eng.forward(batch, w)
        eng.persist_stamps = None
    else:
        eng.forward(batch, w)
        eng.persist_bwd_stamps = st
        eng.loss_and_backward(w)
        eng.persist_bwd_stamps = None
    torch.cuda.synchronize()
    t = st.view(256, 16).double().cpu().numpy() * 0.01 / 801
    print("== %s: frame %.2f us (mean over workgroups), per-workgroup frames %.2f .. %.2f" % (which, t.sum(1).mean(), t.sum(1).min(), t.sum(1).max()))
    idx = order if which == "forward" else list(range(16))
    for i in idx:
        col = t[:, i]
        by_gi = [col[np.arange(256) % 8 == k].mean() for k in range(8)]
        by_gj = [col[np.arange(256) // 8 == k].mean() for k in range(32)]
        print("%-16s mean %.2f  min %.2f  max %.2f | by gi: %s | by gj: min %.2f (gj %d) max %.2f (gj %d)" %
              (names[i] if which == "forward" else "stage %d" % i, col.mean(), col.min(), col.max(), " ".join("%.2f" % v for v in by_gi),
               min(by_gj), int(np.argmin(by_gj)), max(by_gj), int(np.argmax(by_gj))))
