"""Runs the SAME train step (same batch, same masks) N times and reports, per repetition, every gradient / intermediate tensor whose distance to
the first repetition exceeds what the split-K atomics explain (1e-4 of the tensor's maximum).  A tool for hunting intermittent races."""
import os, sys, json
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from test_gpu_model import REF, dims_pair, to_dev, OM, OT
from multi_speaker_tts_amd.engine import TrainEngine

def main(B=4, Te=64, L=800, reps=30, thr=1e-4):
    dev = torch.device("cuda:0")
    pd, od = dims_pair(**REF)
    values = OM.init_params(od, 17)
    batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=17, ragged=True), dev)
    eng = TrainEngine(pd, device=dev, values=values)
    w = eng.plan(B, Te, L)
    names = ["linear", "mel_out", "d_post", "d_linear", "d_pj"] + ["post_dz"]
    def snap():
        eng.forward(batch, w, seed=5)
        eng.loss_and_backward(w)
        torch.cuda.synchronize()
        out = {"grad": eng.params.grad.clone()}
        for n in names:
            t = getattr(w, n)
            if isinstance(t, (list, tuple)):
                for i, x in enumerate(t):
                    out["%s[%d]" % (n, i)] = x.clone()
            else:
                out[n] = t.clone()
        return out
    ref = snap()
    gnames = {k: eng.params.g(k) for k in eng.params.export(grads=True).keys()} if False else None
    for r in range(reps):
        cur = snap()
        bad = {}
        for k in ref:
            a, b = ref[k].double(), cur[k].double()
            e = float((a - b).abs().max() / (a.abs().max() + 1e-30))
            if e > thr:
                bad[k] = e
        if "grad" in bad:
            g0, g1 = ref["grad"], cur["grad"]
            per = {}
            ex0 = eng.params.export(grads=True)
            eng.params.grad.copy_(g0); ex_a = eng.params.export(grads=True)
            eng.params.grad.copy_(g1); ex_b = eng.params.export(grads=True)
            for k in ex_a:
                e = float(np.abs(ex_a[k].astype(np.float64) - ex_b[k]).max() / (np.abs(ex_a[k]).max() + 1e-30))
                if e > thr:
                    per[k] = e
            bad["per_gradient"] = per
        print(json.dumps(dict(rep=r, fallbacks=[eng.persist_fallbacks, eng.persist_bwd_fallbacks, eng.persist_enc_fallbacks], bad=bad)), flush=True)

if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
