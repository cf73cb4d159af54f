"""Generates tests/golden/tiny_train_step.npz from the fp64 oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The TF reference cannot run here (no TensorFlow, no network),
so these are oracle outputs, not reference outputs - parity stays "unpinned" (oracle/__init__.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as OM, train as OT  # noqa: E402

cfg = dict(emb=32, enc_conv_ch=32, enc_lstm=16, spk=16, prenet=16, dec_lstm=32, n_mel=8, post_ch=16,
           bank_ch=8, proj1_ch=16, birnn=8, n_spec=20, spk_lstm=16, att_k=31)
seed, B, Te, L = 5, 3, 10, 6
d = OM.Dims(**cfg)
params = OM.init_params(d, seed)
batch = OT.synthetic_batch(d, B, Te, L, seed=seed, ragged=True)
masks = OT.make_masks(d, B, Te, L + 1, True, seed=OT.step_seed(1234, 0))
_, _, sc, grads, out = OT.train_step(params, None, d, batch, masks, 0, return_grads=True)
k = "decoder/decoder/attention_wrapper/multi_rnn_cell/cell_0/zoneout_lstm_cell/kernel"
# the fixture is self-contained: inputs (variables, batch, keep-masks) and expected outputs, so the GPU test that consumes it
# (tests/test_gpu_model.py::test_golden_fixture_hip) needs nothing from oracle/
inputs = {"p/" + n: np.asarray(v) for n, v in params.items()}
inputs.update({"b/" + n: v.numpy() for n, v in batch.items()})
inputs.update({"m/" + n: v.numpy() for n, v in masks.items()})
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_train_step.npz"),
                    cfg=json.dumps(cfg), seed=seed, B=B, Te=Te, L=L, mel=out["Mel"].numpy(), linear=out["Linear"].numpy(),
                    stop=out["Stop_Logit"].numpy(), align=out["Attention_History"].numpy(), loss=sc["Loss"],
                    scalars=json.dumps({k_: float(v) for k_, v in sc.items()}),
                    grad_cell0=grads[k].numpy(), **{"g/" + n: v.numpy() for n, v in grads.items()}, **inputs)
print("wrote golden; loss", sc["Loss"])
