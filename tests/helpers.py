"""Shared helpers of the parity tests (the oracle is only ever the checker)."""
import numpy as np
import torch

from oracle import model as OM
from multi_speaker_tts_amd.params import Dims


def small_dims(**kw):
    """Reduced widths (attention dims stay at the kernels' built sizes A=128, CH=32)."""
    base = dict(emb=32, enc_conv_ch=32, enc_lstm=16, spk=16, prenet=16, dec_lstm=32, n_mel=8, post_ch=16,
                bank_ch=8, proj1_ch=16, birnn=8, n_spec=20, spk_lstm=16, att_k=31)
    base.update(kw)
    return base


def dims_pair(**kw):
    cfg = small_dims(**kw)
    return Dims(**cfg), OM.Dims(**cfg)


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def to_dev(batch, dev):
    return {k: v.to(dev).contiguous() for k, v in batch.items()}


def t2n(t):
    return t.detach().cpu().numpy()
