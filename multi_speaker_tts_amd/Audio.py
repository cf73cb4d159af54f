"""Drop-in for the reference's ``Audio.melspectrogram`` (Audio.py:29-32) on the MI355X.

Host side only prepares constants (windowed DFT basis, Slaney mel filterbank - both functions of
the hyper parameters, built once in float64 and cached on the device); the waveform -> mel
computation itself is ``mstts_stft_mel`` in libmstts_hip.so.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

from . import lib


def _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate):
    return (num_freq - 1) * 2, int(frame_shift_ms / 1000 * sample_rate), int(frame_length_ms / 1000 * sample_rate)


def _slaney_mel_to_hz(m):
    f_sp, min_hz = 200.0 / 3, 1000.0
    min_mel, step = min_hz / f_sp, np.log(6.4) / 27.0
    m = np.asarray(m, np.float64)
    return np.where(m >= min_mel, min_hz * np.exp(step * (m - min_mel)), f_sp * m)


def _slaney_hz_to_mel(f):
    f_sp, min_hz = 200.0 / 3, 1000.0
    min_mel, step = min_hz / f_sp, np.log(6.4) / 27.0
    f = np.asarray(f, np.float64)
    return np.where(f >= min_hz, min_mel + np.log(np.maximum(f, 1e-10) / min_hz) / step, f / f_sp)


def mel_filterbank(sample_rate, n_fft, n_mels):
    """Triangular Slaney-scale filters with area normalisation: what ``librosa.filters.mel(sr,
    n_fft, n_mels)`` returns with its defaults (the call at Audio.py:84)."""
    freqs = np.linspace(0, sample_rate / 2.0, 1 + n_fft // 2)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(sample_rate / 2.0), n_mels + 2))
    fb = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        fb[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (hi - lo))
    return fb


@functools.lru_cache(maxsize=8)
def _constants(num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, device):
    n_fft, hop, win = _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    nb = (n_fft // 2 + 1 + 3) // 4 * 4
    off = (n_fft - win) // 2                      # the Hann window sits centred in the n_fft frame
    n = np.arange(win)
    hann = 0.5 - 0.5 * np.cos(2 * np.pi * n / win)
    ang = 2 * np.pi * np.outer(n + off, np.arange(n_fft // 2 + 1)) / n_fft
    basis = np.zeros((win, 2 * nb))
    basis[:, : n_fft // 2 + 1] = hann[:, None] * np.cos(ang)
    basis[:, nb: nb + n_fft // 2 + 1] = -hann[:, None] * np.sin(ang)
    fb_t = np.zeros((nb, num_mels))
    fb_t[: n_fft // 2 + 1] = mel_filterbank(sample_rate, n_fft, num_mels).T
    dev = torch.device(device)
    return (n_fft, hop, win, nb, torch.tensor(basis, dtype=torch.float32, device=dev),
            torch.tensor(fb_t, dtype=torch.float32, device=dev))


def melspectrogram(y, num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, max_abs_value=None,
                   spectral_subtract=False, device="cuda", return_tensor=False):
    """Same signature and result layout ([num_mels, frames]) as the reference function."""
    if spectral_subtract:
        raise NotImplementedError("spectral_subtract is never enabled on the reference's TTS path")
    if max_abs_value is None:
        raise NotImplementedError("only the symmetric normalisation used by the TTS path (Max_Abs_Mel) is built")
    n_fft, hop, win, nb, basis, fb_t = _constants(num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, str(device))
    wav = torch.as_tensor(np.asarray(y, dtype=np.float32)).to(basis.device) if not torch.is_tensor(y) else y.to(basis.device, torch.float32)
    wav = wav.contiguous()
    n = wav.numel()
    frames = 1 + n // hop
    L = lib.load()
    ws = torch.empty(int(L.mstts_stft_mel_ws_floats(n, n_fft, frames)), dtype=torch.float32, device=basis.device)
    out = torch.empty(frames, num_mels, dtype=torch.float32, device=basis.device)
    lib.call("mstts_stft_mel", lib.ptr(wav), n, 0.97, lib.ptr(basis), lib.ptr(fb_t), n_fft, hop, win, num_mels,
             float(max_abs_value), lib.ptr(ws), lib.ptr(out), frames)
    out = out.t()
    return out if return_tensor else out.cpu().numpy()
