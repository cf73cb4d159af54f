"""Keep-mask streams of one forward pass (dropout + zoneout), drawn on the device with the
library's Philox4x32-10 kernel.  Replaces the stateful tf.random_uniform draws of
tf.layers.dropout (Modules.py:41-45,137-141,248-253) and ZoneoutLSTMCell.dropout_no_scale
(ZoneoutLSTMCell.py:266-271).  Stream ids are part of the library's specification
(DESIGN.md "Randomness"): the mask of global sample g is the Philox stream (seed, stream id, g); element (o, c) of the
sample's slice is draw o * inner + c of it.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib

STREAM = {"enc_conv_drop": 1, "enc_zc_fw": 10, "enc_zh_fw": 11, "enc_zc_bw": 12, "enc_zh_bw": 13,
          "prenet_drop": 20, "dec_zc": 30, "dec_zh": 31, "post_drop": 40,
          "v_zc_fw": 50, "v_zh_fw": 51, "v_zc_bw": 52, "v_zh_bw": 53, "s_zc": 60, "s_zh": 61}


def batch_axis(name):
    """Conv-block dropout masks are batch-major [B, T, C] (samples on axis 0); every other mask is step-major [S, B, C]."""
    return 0 if name.startswith(("enc_conv_drop", "post_drop")) else 1


def step_seed(base_seed, step):
    return (int(base_seed) + 1000003 * int(step)) & 0xFFFFFFFFFFFFFFFF


def table(d, B, T_enc, S, training, speaker_windows=0, vocoder=False):
    """[(name, stream id, shape, keep probability)]"""
    t = [("prenet_drop_%d" % i, STREAM["prenet_drop"] + i, (S, B, d.prenet), 1 - d.prenet_drop) for i in range(d.prenet_n)]
    if not training:
        return t
    for i in range(d.enc_conv_n):
        t.append(("enc_conv_drop_%d" % i, STREAM["enc_conv_drop"] + i, (B, T_enc, d.enc_conv_ch), 1 - d.conv_drop))
    for dr in ("fw", "bw"):
        for k in ("zc", "zh"):
            t.append(("enc_%s_%s" % (k, dr), STREAM["enc_%s_%s" % (k, dr)], (T_enc, B, d.enc_lstm), 1 - d.zoneout))
    for l in range(d.dec_lstm_n):
        for k in ("zc", "zh"):
            t.append(("dec_%s_%d" % (k, l), STREAM["dec_" + k] + 2 * l, (S, B, d.dec_lstm), 1 - d.zoneout))
    for i in range(d.post_n):
        cout = d.post_ch if i < d.post_n - 1 else d.n_mel
        t.append(("post_drop_%d" % i, STREAM["post_drop"] + i, (B, S, cout), 1 - d.conv_drop))
    if vocoder:
        for dr in ("fw", "bw"):
            for k in ("zc", "zh"):
                t.append(("v_%s_%s" % (k, dr), STREAM["v_%s_%s" % (k, dr)], (S, B, d.birnn), 1 - d.zoneout))
    if speaker_windows:
        for i in range(d.spk_lstm_n):
            for k in ("zc", "zh"):
                t.append(("s_%s_%d" % (k, i), STREAM["s_" + k] + 2 * i, (d.spk_frames, speaker_windows, d.spk_lstm), 1 - d.zoneout))
    return t


class MaskSet:
    """Preallocated uint8 mask buffers for one shape; ``draw`` refills them for a seed."""

    def __init__(self, d, B, T_enc, S, training, device, rank=0, alloc=None, **kw):
        """alloc: optional callable(n_bytes) -> uint8 tensor of that length (or None while a caller only counts bytes): the train engine
        carves the masks from its workspace arena instead of allocating them per shape."""
        self.spec = table(d, B, T_enc, S, training, **kw)
        self.rank = rank
        self.buf = {}
        for name, _, shape, _ in self.spec:
            n = int(np.prod(shape))
            if alloc is not None:
                t = alloc((n + 3) // 4 * 4)
                self.buf[name] = None if t is None else t[:n].view(shape)
            else:
                self.buf[name] = torch.empty((n + 3) // 4 * 4, dtype=torch.uint8, device=device)[:n].view(shape)

    def draw(self, seed, only=None):
        """Masks are keyed by (seed, stream, GLOBAL sample index, position in the sample): rank r holds the samples
        r * B .. r * B + B - 1 of the global batch, so 1/2/4/8-rank runs draw the same mask for the same sample.
        only: a predicate on the mask name (draw a subset now, the rest later)."""
        for name, stream, shape, keep in self.spec:
            if only is not None and not only(name):
                continue
            ax = batch_axis(name)
            outer, nb = (1, shape[0]) if ax == 0 else (shape[0], shape[1])
            inner = int(np.prod(shape[ax + 1:]))
            lib.call("mstts_philox_keep_mask_rows", lib.ptr(self.buf[name]), outer, nb, inner, seed, stream, self.rank * nb, float(keep))

    def load(self, masks):
        """Inject externally supplied masks (tests)."""
        for name, _, shape, _ in self.spec:
            if name in masks:
                self.buf[name].copy_(torch.as_tensor(masks[name]).to(torch.uint8).reshape(shape))

    def __getitem__(self, name):
        return self.buf[name]

    def get(self, name):
        return self.buf.get(name)
