import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from tests.test_gpu_persist import _engine, BWD, _bwd_snapshot
from tests.helpers import to_dev, rel_err, t2n
from oracle import train as OT
dev = torch.device("cuda:0")
eng, od = _engine(dev)
B, Te, L = 32, 256, 6
batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
w = eng.plan(B, Te, L)
eng.forward(batch, w, seed=OT.step_seed(1234, 0))
snaps = []
for rep in range(4):
    for k in BWD: getattr(w, k).zero_()
    w.d_in0.zero_()
    eng.loss_and_backward(w)
    torch.cuda.synchronize()
    snaps.append(_bwd_snapshot(eng, w))
w.persist_bwd = False
for k in BWD: getattr(w, k).zero_()
w.d_in0.zero_()
eng.loss_and_backward(w); torch.cuda.synchronize()
b = _bwd_snapshot(eng, w)
print("token lengths", batch["Token_Length"].cpu().numpy())
for i, a in enumerate(snaps):
    print("rep", i, {k: "%.2e" % rel_err(a[k], b[k]) for k in ("dq_hist", "dg1", "dg0", "de_hist", "d_ctx")}, "vs rep0 dq %.2e dg1 %.2e" % (rel_err(a["dq_hist"], snaps[0]["dq_hist"]), rel_err(a["dg1"], snaps[0]["dg1"])))
    d = np.abs(a["dq_hist"].astype(np.float64) - b["dq_hist"]) / np.abs(b["dq_hist"]).max()
    idx = np.argwhere(d > 2e-5)
    if len(idx):
        print("   differing (step,row,unit) count", len(idx), "steps", sorted(set(idx[:, 0])), "rows", sorted(set(idx[:, 1]))[:40], "units", sorted(set(idx[:, 2]))[:40])
