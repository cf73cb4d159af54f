"""WaveGlow vocoder, inference direction, on the MI355X (reference: WaveGlow/Modules.py:177-208,210-327,354-371,
WaveGlow/Inv1x1.py:9-41; wired at MSTTS_SV.py:117-125,325-389).

Python owns buffers and the schedule; every arithmetic step is a libmstts_hip.so call:
  * the transposed-conv mel upsampler = one GEMM (frame x all taps) + mstts_wg_overlap_add,
  * per coupling layer: the initial 1x1 conv, ONE GEMM for the conditioning 1x1 convs of all WaveNet layers, then per
    layer the dilated K=3 conv as an implicit-im2col GEMM (win_dil) accumulated onto its conditioning block,
    mstts_wg_gate, the res/skip 1x1 GEMM, mstts_wg_res_skip; the zero-initialised output conv; mstts_wg_coupling_inv
    (affine inverse + inverse 1x1 conv + early-latent re-injection),
  * tf.random.normal -> mstts_philox_normal (or injected arrays, which is what the parity tests use).
Weight normalisation (g * v / ||v||, WaveGlow/Modules.py:9-34) and the matrix inverses of the 1x1 kernels are
constants of a checkpoint and are folded once at load time on the host.
"""
from __future__ import annotations

from dataclasses import dataclass
import math
import os

import numpy as np
import torch

from . import lib
from .lib import call, gemm, ptr

P_WG = "waveglow/"


@dataclass
class WGDims:
    """hp.WaveGlow / hp.Sound.Mel_Dim (Hyper_Parameters.py:196-210)."""
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
        return self.groups - (flow // self.early_every) * self.early_size

    @property
    def z_channels(self):
        return self.groups - (int(math.ceil(self.flows / self.early_every)) - 1) * self.early_size

    @classmethod
    def from_hp(cls, hp):
        w = hp.WaveGlow
        return cls(n_mel=hp.Sound.Mel_Dim, flows=w.Flows, groups=w.Groups, early_every=w.Early_Every, early_size=w.Early_Size,
                   up_k=w.Upsample.Kernel_Size, up_stride=w.Upsample.Strides, layers=w.WaveNet.Layers, ch=w.WaveNet.Channels,
                   k=w.WaveNet.Kernel_Size)


def variable_table(d: WGDims):
    """[(name, shape)] in the reference's variable scopes (inferred; unverified against a real checkpoint)."""
    t = [(P_WG + "conv2d_transpose/kernel", (1, d.up_k, d.n_mel, d.n_mel)), (P_WG + "conv2d_transpose/bias", (d.n_mel,))]
    cm = d.groups * d.n_mel
    for f in range(d.flows):
        c = d.channels(f)
        p = P_WG + "affine_coupling_layer_%d/" % f
        t.append((p + "invertible_1x1/kernel", (c, c)))

        def wn(name, k, cin, cout):
            t.extend([(p + "wavenet/" + name + "/g", (cout,)), (p + "wavenet/" + name + "/kernel", (1, k, cin, cout)),
                      (p + "wavenet/" + name + "/bias", (cout,))])
        wn("audio_initial_conv", 1, c // 2, d.ch)
        for i in range(d.layers):
            wn("audio_in_%d" % i, d.k, d.ch, 2 * d.ch)
            wn("mel_cond_%d" % i, 1, cm, 2 * d.ch)
            wn("res_%d" % i, 1, d.ch, 2 * d.ch if i < d.layers - 1 else d.ch)
        t.append((p + "wavenet/conv1d/kernel", (1, d.ch, c)))
        t.append((p + "wavenet/conv1d/bias", (c,)))
    return t


def random_values(d: WGDims, seed=0):
    """The reference's initialisers (uniform(0, .02) upsampler, glorot weight-norm kernels, N(0,1) 1x1 kernels with
    positive determinant, zero output conv).  A WaveGlow initialised like this is the identity coupling."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in variable_table(d):
        if name.endswith("bias") or "wavenet/conv1d" in name:
            v = np.zeros(shape)
        elif "conv2d_transpose/kernel" in name:
            v = g.uniform(0, 0.02, shape)
        elif "invertible_1x1" in name:
            v = g.normal(0, 1, shape)
            if np.linalg.det(v) < 0:
                v[:, 0] *= -1
        elif name.endswith("/g"):
            v = g.uniform(-1, 1, shape) * math.sqrt(6.0 / (2 * shape[0]))
        else:
            rf = int(np.prod(shape[:-2]))
            lim = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            v = g.uniform(-lim, lim, shape)
        out[name] = v.astype(np.float32)
    return out


def _weight_norm(g, v):
    v = np.asarray(v, np.float64)
    ss = (v * v).sum(axis=(0, 1, 2), keepdims=True)
    return np.asarray(g, np.float64) * v / np.sqrt(np.maximum(ss, 1e-5))


def _conv_two_pieces(rows, n_out):
    """Whether the dilated convolution [rows, k ch] x [k ch, n_out] runs as two reduction pieces onto a zeroed buffer (see WaveGlowEngine.infer).
    Fitted to tools/wg_conv_ab.py at the reference widths (per-layer microseconds, one piece | two pieces: batch 1: 92 | 64, 2: 99 | 118,
    4: 185 | 164, 5: 187 | 174, 6: 277 | 322, 8: 268 | 330, 12: 534 | 501, 16: 571 | 518, 24: 838 | 853, 32: 923 | 1050): the 128 x 128-tile kernel's
    time steps at 256, 512 and then every 1 024 tiles, the 256 x 256-tile kernel's at every 256 workgroups; in units of the latter's round."""
    cols128, cols256 = -(-n_out // 128), -(-n_out // 256)
    t128 = -(-rows // 128) * cols128
    w256 = -(-rows // 256) * cols256 * 2
    if t128 <= 128:                      # a fraction of the chip either way: two pieces of the small kernel fill it better
        return True
    if w256 < 160:                       # (below the library's threshold for the big tile)
        return False
    one = 0.58 if t128 <= 256 else 1.12 if t128 <= 512 else 1.65 * -(-t128 // 1024)
    return -(-w256 // 256) < one


class WaveGlowEngine:
    def __init__(self, dims: WGDims = None, device="cuda", values=None, seed=1234):
        self.d = dims or WGDims()
        self.device = torch.device(device)
        self.seed = seed
        lib.load()
        vals = values if values is not None else random_values(self.d, seed)
        missing = [n for n, _ in variable_table(self.d) if n not in vals]
        if missing:
            raise ValueError("WaveGlow variables missing: %s ..." % missing[:3])
        self.load(vals)
        self._keep = []
        self.split_in = 0          # > 1: split the dilated-conv GEMM's reduction (atomic accumulation); 0 / 1 = off (see infer)
        self.conv_two_pieces = os.environ.get("MSTTS_WG_TWO_PIECES", "1") != "0"     # the dilated convolution as two reduction pieces onto a zeroed buffer (see infer)

    # ------------------------------------------------------------------ checkpoint constants, folded on the host once
    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def load(self, v):
        d = self.d
        # conv2d_transpose kernel [1,K,Cout,Cin] -> [Cin, K*Cout]: one GEMM gives every tap product of a frame
        wt = np.asarray(v[P_WG + "conv2d_transpose/kernel"], np.float64)[0]
        self.up_w = self._dev(wt.transpose(2, 0, 1).reshape(d.n_mel, d.up_k * d.n_mel))
        self.up_b = self._dev(v[P_WG + "conv2d_transpose/bias"])
        self.flow = []
        for f in range(d.flows):
            c = d.channels(f)
            p = P_WG + "affine_coupling_layer_%d/wavenet/" % f
            wn = lambda n: _weight_norm(v[p + n + "/g"], v[p + n + "/kernel"])[0]
            F = {"c": c}
            F["w_init"], F["b_init"] = self._dev(wn("audio_initial_conv")[0]), self._dev(v[p + "audio_initial_conv/bias"])
            # the conditioning convs of all layers side by side; the bias of each layer's dilated conv rides along
            F["w_cond"] = self._dev(np.concatenate([wn("mel_cond_%d" % i)[0] for i in range(d.layers)], axis=1))
            F["b_cond"] = self._dev(np.concatenate([np.asarray(v[p + "mel_cond_%d/bias" % i], np.float64) + np.asarray(v[p + "audio_in_%d/bias" % i], np.float64)
                                                    for i in range(d.layers)]))
            F["w_in"] = [self._dev(wn("audio_in_%d" % i).reshape(d.k * d.ch, 2 * d.ch)) for i in range(d.layers)]
            F["w_res"] = [self._dev(wn("res_%d" % i)[0]) for i in range(d.layers)]
            F["b_res"] = [self._dev(v[p + "res_%d/bias" % i]) for i in range(d.layers)]
            F["w_out"], F["b_out"] = self._dev(np.asarray(v[p + "conv1d/kernel"])[0]), self._dev(v[p + "conv1d/bias"])
            F["w_inv"] = self._dev(np.linalg.inv(np.asarray(v[P_WG + "affine_coupling_layer_%d/invertible_1x1/kernel" % f], np.float64)))
            self.flow.append(F)

    def _f(self, *shape):
        n = int(np.prod(shape))
        t = torch.empty((n + 3) // 4 * 4, dtype=torch.float32, device=self.device)[:n].view(shape)
        self._keep.append(t)
        return t

    # ------------------------------------------------------------------ Glow_Inference
    @lib.deterministic_gemm()          # bit-reproducible per latent seed: no K-cuts with atomics inside the flow's contractions
    def infer(self, mel, noise=None, seed=None, sigma=1.0):
        """mel [N, T, n_mel] (tensor or array) -> wav tensor [N, (T-1)*stride + kernel].  noise: {"z": [N,L/G,z_channels],
        "early_<flow>": [N,L/G,early_size]} to inject the latents, else Philox normals of `seed`."""
        d = self.d
        self._keep = []
        mel = (mel if torch.is_tensor(mel) else torch.from_numpy(np.asarray(mel))).to(self.device, torch.float32).contiguous()
        N, T, C = mel.shape
        assert C == d.n_mel
        L = (T - 1) * d.up_stride + d.up_k
        if L % d.groups:
            raise ValueError("upsampled length %d is not a multiple of Groups=%d" % (L, d.groups))
        Lg, rows, cm, ch = L // d.groups, N * (L // d.groups), d.groups * C, d.ch
        # Upsample_Mel: every tap product of every frame, then overlap-add (+ bias)
        Y = self._f(N * T, d.up_k * C)
        gemm(mel, self.up_w, Y, N * T, d.up_k * C, C, C, d.up_k * C, d.up_k * C)
        up = self._f(N, L, C)
        call("mstts_wg_overlap_add", ptr(Y), ptr(self.up_b), ptr(up), N, T, d.up_k, d.up_stride, C)
        melg = up                                     # viewed as [rows, G*C]: contiguous regrouping (:180-187)
        seed = self.seed if seed is None else seed

        def latent(key, width, stream, scale):
            t = self._f(rows, width)
            if noise is not None:
                t.copy_((noise[key] if torch.is_tensor(noise[key]) else torch.from_numpy(np.asarray(noise[key]))).to(self.device, torch.float32).reshape(rows, width))
            else:
                call("mstts_philox_normal", ptr(t), rows * width, seed, stream, scale)
            return t
        audio = latent("z", d.z_channels, 70, 1.0)
        x, z, out = self._f(rows, ch), self._f(rows, ch), self._f(rows, ch)
        cond, rs = self._f(rows, d.layers * 2 * ch), self._f(rows, 2 * ch)
        ldc = d.layers * 2 * ch
        # The [rows, 3*ch] x [3*ch, 2*ch] conv GEMM has only ceil(rows/128) * (2*ch/128) = 344 output tiles of 128 rows at batch 4 x 40
        # frames on 256 CUs; mstts_gemm_f32 switches to 64-row tiles for such shapes (688 tiles), which beats the split-K form
        # used before (33.0 vs 33.6 ms per batch) and keeps the launch free of atomics - results are bit-reproducible run to run.
        # Round 5: where it pays, the convolution's output goes to its own ZEROED buffer as exactly two reduction pieces (split_k = 2).  0 + p + q is
        # the same float whichever piece's atomic lands first (fp addition commutes; three pieces would not be order-free), so the flow is still
        # bit-reproducible per latent seed - and at batch 4 x 40 frames 22 x 4 tiles x 2 pieces = 176 workgroups reach the 256 x 256-tile kernel
        # (csrc/gemm_split.inc), where the 128 x 128 one runs 344 tiles as a round and a third: 185 -> 164 us per layer, gate included (it adds the
        # two buffers).  _conv_two_pieces: a fixed function of the shape (tools/wg_conv_ab.py), so a seed gives the same samples in every process.
        split_in = self.split_in or 1
        two = self.conv_two_pieces and split_in == 1 and ch % 4 == 0 and _conv_two_pieces(rows, 2 * ch)
        conv = self._f(rows, 2 * ch) if two else None
        for f in reversed(range(d.flows)):
            F = self.flow[f]
            c = F["c"]
            h = c // 2
            gemm(audio, F["w_init"], x, rows, ch, h, c, ch, ch, bias=F["b_init"])                       # audio0 = audio[:, :h]
            gemm(melg, F["w_cond"], cond, rows, ldc, cm, cm, ldc, ldc, bias=F["b_cond"])
            for i in range(d.layers):
                last = i == d.layers - 1
                if two:
                    conv.zero_()
                    gemm(x, F["w_in"][i], conv, rows, 2 * ch, d.k * ch, ch, 2 * ch, 2 * ch, split_k=2, win=(Lg, ch, (d.k - 1) // 2, 2 ** i))
                    call("mstts_wg_gate_add", ptr(cond, i * 2 * ch), ldc, ptr(conv), ptr(z), rows, ch)
                else:
                    gemm(x, F["w_in"][i], cond, rows, 2 * ch, d.k * ch, ch, 2 * ch, ldc, accumulate=True, split_k=split_in,
                         win=(Lg, ch, (d.k - 1) // 2, 2 ** i), c_off=i * 2 * ch)
                    call("mstts_wg_gate", ptr(cond, i * 2 * ch), ldc, ptr(z), rows, ch)
                nres = ch if last else 2 * ch
                gemm(z, F["w_res"][i], rs, rows, nres, ch, ch, nres, nres, bias=F["b_res"][i])
                call("mstts_wg_res_skip", ptr(z), ptr(rs), ptr(x), ptr(out), rows, ch, int(last), int(i == 0))
            ls_b = self._f(rows, c)
            gemm(out, F["w_out"], ls_b, rows, c, ch, ch, c, c, bias=F["b_out"])
            early = f % d.early_every == 0 and f > 0
            ce = d.early_size if early else 0
            e = latent("early_%d" % f, ce, 71 + f, 1.0) if early else None
            nxt = self._f(rows, c + ce)
            call("mstts_wg_coupling_inv", ptr(audio), ptr(ls_b), ptr(F["w_inv"]), ptr(e) if early else None, float(sigma), ptr(nxt), rows, c, ce)
            audio = nxt
        return audio.view(N, Lg * d.groups)


# ---- MSTTS_SV.Inference_WaveGlow host logic (MSTTS_SV.py:335-375,452-458) ----------------------------------------------
def split_mels(mels, split):
    """Every mel cut into `split`-frame chunks; index[i] = (first, last+1) chunk of utterance i."""
    chunks, index = [], []
    for mel in mels:
        parts = [mel[x:x + split] for x in range(0, mel.shape[0], split)]
        start = index[-1][1] if index else 0
        chunks.extend(parts)
        index.append((start, start + len(parts)))
    return chunks, index


def vocode(engine: WaveGlowEngine, mels, split, batch, noise_seed=None):
    """Chunk, zero-pad to the longest chunk, run `batch` chunks at a time, stitch (the per-chunk tail of kernel - stride
    samples is NOT trimmed, as in the reference: WaveGlow/Modules.py:179 is commented out)."""
    chunks, index = split_mels(mels, split)
    tmax = max(c.shape[0] for c in chunks)
    pat = np.zeros((len(chunks), tmax, engine.d.n_mel), np.float32)
    for i, c in enumerate(chunks):
        pat[i, :c.shape[0]] = c
    wavs = []
    for b0 in range(0, len(chunks), batch):
        seed = None if noise_seed is None else noise_seed + b0
        wavs.append(engine.infer(pat[b0:b0 + batch], seed=seed).cpu().numpy())
    allw = np.concatenate(wavs, axis=0)
    return [allw[a:b].reshape(-1) for a, b in index]


def export_length(stop, frame_shift_ms, sample_rate):
    stop = np.asarray(stop)
    cut = int(np.argmax(stop > 0.5)) if (stop > 0.5).any() else stop.shape[0]
    return int(cut * frame_shift_ms / 1000 * sample_rate)
