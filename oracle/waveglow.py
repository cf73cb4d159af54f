"""torch-CPU restatement of the reference's WaveGlow vocoder, inference direction (TEST INFRASTRUCTURE - see __init__).

PARITY UNPINNED: the reference (TensorFlow 1.x) cannot run here and ships no vectors for this path; this file follows
WaveGlow/Modules.py and WaveGlow/Inv1x1.py line by line and is pinned only by its own invariants (the flow is
invertible: `glow_forward` followed by `glow_inference` on the emitted latents returns the audio).

Layouts are the reference's: activations [N, T, C]; weight-norm kernels [1, K, Cin, Cout] with gain g[Cout]; the
transposed-conv kernel [1, K, Cout, Cin]; invertible 1x1 kernels [C, C] applied as x @ W.
All randomness (the latent z and the re-injected early outputs) enters through explicit arrays.
"""
from __future__ import annotations

from dataclasses import dataclass
import math
import numpy as np
import torch
import torch.nn.functional as F

P_WG = "waveglow/"


@dataclass
class WGDims:
    """hp.WaveGlow / hp.Sound (Hyper_Parameters.py:196-210)."""
    n_mel: int = 80
    flows: int = 12
    groups: int = 8
    early_every: int = 4
    early_size: int = 2
    up_k: int = 1024
    up_stride: int = 256
    layers: int = 8
    ch: int = 512
    k: int = 3

    def channels(self, flow):
        """Audio channels seen by coupling layer `flow` (Glow_Train order: early outputs split off every early_every)."""
        return self.groups - (flow // self.early_every) * self.early_size

    @property
    def z_channels(self):
        """Restructure_Inference_Data: Groups - (ceil(Flows / Early_Every) - 1) * Early_Size (WaveGlow/Modules.py:189-193)."""
        return self.groups - (int(math.ceil(self.flows / self.early_every)) - 1) * self.early_size


def param_specs(d: WGDims):
    """[(name, shape, init)], names follow the reference's variable scopes (inferred, unverified against a checkpoint)."""
    s = [(P_WG + "conv2d_transpose/kernel", (1, d.up_k, d.n_mel, d.n_mel), "uniform:0:0.02"),      # Upsample_Mel :198-208
         (P_WG + "conv2d_transpose/bias", (d.n_mel,), "zeros")]
    cm = d.groups * d.n_mel
    for f in range(d.flows):
        c = d.channels(f)
        p = P_WG + "affine_coupling_layer_%d/" % f
        s.append((p + "invertible_1x1/kernel", (c, c), "inv1x1"))                                   # Inv1x1.py:13-18

        def wn(name, k, cin, cout):
            s.extend([(p + "wavenet/" + name + "/g", (cout,), "glorot1"), (p + "wavenet/" + name + "/kernel", (1, k, cin, cout), "glorot"),
                      (p + "wavenet/" + name + "/bias", (cout,), "zeros")])
        wn("audio_initial_conv", 1, c // 2, d.ch)                                                   # :257-264
        for i in range(d.layers):
            wn("audio_in_%d" % i, d.k, d.ch, 2 * d.ch)                                              # :267-275
            wn("mel_cond_%d" % i, 1, cm, 2 * d.ch)                                                  # :276-284
            wn("res_%d" % i, 1, d.ch, 2 * d.ch if i < d.layers - 1 else d.ch)                       # :293-300
        s.append((p + "wavenet/conv1d/kernel", (1, d.ch, c), "zeros"))                              # :314-320 (zero-initialised)
        s.append((p + "wavenet/conv1d/bias", (c,), "zeros"))
    return s


def init_params(d: WGDims, seed=0, trained_like=True):
    """Random variables.  With `trained_like` the zero-initialised output conv gets small random values so that the
    coupling is not the identity (a freshly initialised WaveGlow would make every parity test vacuous)."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape, init in param_specs(d):
        if init == "zeros":
            v = np.zeros(shape)
            if trained_like and "wavenet/conv1d" in name:
                v = g.normal(0, 0.05, shape)
        elif init.startswith("uniform"):
            _, lo, hi = init.split(":")
            v = g.uniform(float(lo), float(hi), shape)
        elif init == "inv1x1":
            v = g.normal(0, 1, shape)
            if np.linalg.det(v) < 0:
                v[:, 0] *= -1
        elif init == "glorot1":
            v = g.uniform(0.5, 1.5, shape)
        else:
            rf = int(np.prod(shape[:-2]))
            lim = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            v = g.uniform(-lim, lim, shape)
        out[name] = v.astype(np.float64)
    return out


def weight_norm(g, v):
    """Get_Weight_Norm_Variable (WaveGlow/Modules.py:9-34): g * l2_normalize(v, axis=[0,1,2], epsilon=1e-5)."""
    ss = (v * v).sum(dim=(0, 1, 2), keepdim=True)
    return g * v * torch.rsqrt(torch.clamp(ss, min=1e-5))


def wn_conv1d(p, prefix, x, dilation=1):
    """Weight_Norm_Conv1D, padding 'same', stride 1 (WaveGlow/Modules.py:85-135): cross-correlation, NWC."""
    w = weight_norm(p[prefix + "/g"], p[prefix + "/kernel"])[0]            # [K, Cin, Cout]
    K = w.shape[0]
    total = (K - 1) * dilation
    left = total // 2
    y = F.conv1d(F.pad(x.transpose(1, 2), (left, total - left)), w.permute(2, 1, 0), dilation=dilation)
    return y.transpose(1, 2) + p[prefix + "/bias"]


def upsample_mel(p, d: WGDims, mel):
    """Upsample_Mel (WaveGlow/Modules.py:198-208): conv2d_transpose, kernel (1, K), stride (1, S), padding VALID ->
    length (T - 1) * S + K."""
    w = p[P_WG + "conv2d_transpose/kernel"][0]                             # [K, Cout, Cin]
    y = F.conv_transpose1d(mel.transpose(1, 2), w.permute(2, 1, 0), stride=d.up_stride)
    return y.transpose(1, 2) + p[P_WG + "conv2d_transpose/bias"]


def restructure_inference_mel(p, d: WGDims, mel):
    """Restructure_Inference_Data (WaveGlow/Modules.py:177-187): upsample, then fold `groups` samples into channels."""
    up = upsample_mel(p, d, mel)
    N, L, C = up.shape
    return up[:, :L // d.groups * d.groups].reshape(N, L // d.groups, d.groups * C)


def wavenet(p, d: WGDims, prefix, audio, mel):
    """WaveNet (WaveGlow/Modules.py:252-327).  Quirk kept: the residual is added to the GATED activation (:309), not
    to the layer input, and the last layer's res conv has `ch` filters that all go to the skip sum."""
    x = wn_conv1d(p, prefix + "audio_initial_conv", audio)
    out = 0
    for i in range(d.layers):
        a = wn_conv1d(p, prefix + "audio_in_%d" % i, x, dilation=2 ** i) + wn_conv1d(p, prefix + "mel_cond_%d" % i, mel)
        t, s = a.chunk(2, dim=-1)
        x = torch.tanh(t) * torch.sigmoid(s)
        rs = wn_conv1d(p, prefix + "res_%d" % i, x)
        if i < d.layers - 1:
            res, skip = rs.chunk(2, dim=-1)
            x = x + res
        else:
            skip = rs
        out = out + skip
    y = out @ p[prefix + "conv1d/kernel"][0] + p[prefix + "conv1d/bias"]
    return y.chunk(2, dim=-1)            # log_s, bias


def coupling_reverse(p, d: WGDims, flow, audio, mel):
    """Affine_Coupling_Layer, reverse=True (WaveGlow/Modules.py:210-250) + Inv1x1 reverse (Inv1x1.py:30-32)."""
    pre = P_WG + "affine_coupling_layer_%d/" % flow
    a0, a1 = audio.chunk(2, dim=-1)
    log_s, b = wavenet(p, d, pre + "wavenet/", a0, mel)
    a1 = (a1 - b) / torch.exp(log_s)
    return torch.cat([a0, a1], dim=-1) @ torch.linalg.inv(p[pre + "invertible_1x1/kernel"])


def coupling_forward(p, d: WGDims, flow, audio, mel):
    pre = P_WG + "affine_coupling_layer_%d/" % flow
    audio = audio @ p[pre + "invertible_1x1/kernel"]
    a0, a1 = audio.chunk(2, dim=-1)
    log_s, b = wavenet(p, d, pre + "wavenet/", a0, mel)
    log_s = torch.clamp(log_s, max=8.0)
    return torch.cat([a0, torch.exp(log_s) * a1 + b], dim=-1)


def glow_inference(p, d: WGDims, mel, noise, sigma=1.0):
    """Glow_Inference (WaveGlow/Modules.py:354-371).  noise: {"z": [N, L/G, z_channels], "early_<flow>": [N, L/G, early_size]}
    (the reference draws them with tf.random.normal; the initial z is NOT scaled by sigma, the re-injected ones are)."""
    melg = restructure_inference_mel(p, d, mel)
    audio = noise["z"]
    for flow in reversed(range(d.flows)):
        audio = coupling_reverse(p, d, flow, audio, melg)
        if flow % d.early_every == 0 and flow > 0:
            audio = torch.cat([noise["early_%d" % flow] * sigma, audio], dim=-1)
    return audio.reshape(audio.shape[0], -1)


def glow_forward(p, d: WGDims, audio, melg):
    """Glow_Train (WaveGlow/Modules.py:329-352) on pre-grouped tensors; returns the latents in the form glow_inference
    consumes, so that glow_inference(glow_forward(x)) == x (used only to pin the restatement)."""
    noise = {}
    for flow in range(d.flows):
        if flow % d.early_every == 0 and flow > 0:
            noise["early_%d" % flow] = audio[:, :, :d.early_size]
            audio = audio[:, :, d.early_size:]
        audio = coupling_forward(p, d, flow, audio, melg)
    noise["z"] = audio
    return noise


def make_noise(d: WGDims, N, Lg, seed):
    g = np.random.default_rng(seed)
    noise = {"z": g.normal(0, 1, (N, Lg, d.z_channels))}
    for flow in range(d.flows):
        if flow % d.early_every == 0 and flow > 0:
            noise["early_%d" % flow] = g.normal(0, 1, (N, Lg, d.early_size))
    return noise


def to_torch(values, dtype=torch.float64):
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in values.items()}


# ---- MSTTS_SV.Inference_WaveGlow chunking / stitching (MSTTS_SV.py:335-375,458) --------------------------------------
def split_mels(mels, split):
    chunks, index = [], []
    for mel in mels:
        parts = [mel[x:x + split] for x in range(0, mel.shape[0], split)]
        start = index[-1][1] if index else 0
        chunks.extend(parts)
        index.append((start, start + len(parts)))
    return chunks, index


def export_length(stop, frame_shift_ms, sample_rate):
    """Export_Inference_WaveGlow (MSTTS_SV.py:452-458): cut at the first stop > 0.5, in samples."""
    stop = np.asarray(stop)
    cut = int(np.argmax(stop > 0.5)) if (stop > 0.5).any() else stop.shape[0]
    return int(cut * frame_shift_ms / 1000 * sample_rate)
