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


@functools.lru_cache(maxsize=8)
def _fft_constants(num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, device):
    """Tables of the one-launch FFT path (mstts_stft_fft): periodic Hann window, e^{-2 pi i k / n_fft} twiddles, the mel filterbank
    row-major with each filter's non-zero bin range - all built in float64."""
    n_fft, hop, win = _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    dev = torch.device(device)
    n = np.arange(win)
    hann = 0.5 - 0.5 * np.cos(2 * np.pi * n / win)
    k = np.arange(n_fft)
    tw = np.stack([np.cos(2 * np.pi * k / n_fft), -np.sin(2 * np.pi * k / n_fft)], axis=1)
    fb = mel_filterbank(sample_rate, n_fft, num_mels).astype(np.float32)
    rng = np.zeros((num_mels, 2), np.int32)
    for c in range(num_mels):
        nz = np.nonzero(fb[c])[0]
        if len(nz):
            rng[c] = (nz[0], nz[-1] + 1)
    t = lambda a, dt: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    return n_fft, hop, win, t(hann, torch.float32), t(tw, torch.float32), t(fb, torch.float32), t(rng, torch.int32)


def stft_features(wavs, num_freq, frame_shift_ms, frame_length_ms, sample_rate, num_mels=None, max_abs_value=4, ref_level_db=20,
                  want_mel=True, want_spec=False, spectral_subtract=False, device="cuda"):
    """Mel and / or linear spectrogram features of SEVERAL waveforms in one kernel launch (mstts_stft_fft): returns a list of
    (mel [frames, num_mels] or None, spec [frames, num_freq] or None) device tensors, one pair per waveform.
    max_abs_value None: the mel is normalised to [0, 1] (Audio._normalize) instead of symmetrically.  spectral_subtract
    (Audio.py:45-46): a first launch leaves the raw magnitudes, a tenth of each waveform's per-bin time mean is subtracted
    (clipped at 0) by a second launch that finishes the features - three launches per waveform instead of one for the batch."""
    n_fft, hop, win, hann, tw, fb, rng = _fft_constants(num_freq, frame_shift_ms, frame_length_ms, num_mels or 1, sample_rate, str(device))
    dev = hann.device
    ws = [torch.as_tensor(np.asarray(y, dtype=np.float32)) if not torch.is_tensor(y) else y.to(torch.float32).reshape(-1) for y in wavs]
    lens = [int(w.numel()) for w in ws]
    if any(n <= n_fft // 2 for n in lens):
        raise ValueError("waveform shorter than the STFT's reflect padding (%d samples)" % (n_fft // 2))
    frames = [1 + n // hop for n in lens]
    woff = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=dev)
    foff = torch.tensor(np.concatenate([[0], np.cumsum(frames)]), dtype=torch.int64, device=dev)
    cat = torch.cat([w.to(dev) for w in ws]).contiguous()
    total, nb = int(sum(frames)), n_fft // 2 + 1
    mel = torch.empty(total, num_mels, dtype=torch.float32, device=dev) if want_mel else None
    spec = torch.empty(total, nb, dtype=torch.float32, device=dev) if want_spec else None
    flags = 1 if max_abs_value is None else 0
    mabs = float(max_abs_value) if max_abs_value is not None else 1.0
    common = (lib.ptr(fb) if want_mel else None, lib.ptr(rng) if want_mel else None, n_fft, hop, win, int(num_mels or 0), mabs, float(ref_level_db))
    if not spectral_subtract:
        lib.call("mstts_stft_fft", lib.ptr(cat), lib.ptr(woff), lib.ptr(foff), len(ws), 0.97, lib.ptr(hann), lib.ptr(tw), *common,
                 lib.ptr(mel) if want_mel else None, lib.ptr(spec) if want_spec else None, total, None, None, 0.0, flags)
    else:
        mag = torch.empty(total, nb, dtype=torch.float32, device=dev)
        lib.call("mstts_stft_fft", lib.ptr(cat), lib.ptr(woff), lib.ptr(foff), len(ws), 0.97, lib.ptr(hann), lib.ptr(tw), None, None,
                 n_fft, hop, win, 0, mabs, float(ref_level_db), None, lib.ptr(mag), total, None, None, 0.0, 2)
        sub = torch.empty(nb, dtype=torch.float32, device=dev)
        one = torch.tensor([0, 0], dtype=torch.int64, device=dev)
        f0 = 0
        for fr in frames:                                   # the mean runs over one waveform's frames
            lib.call("mstts_colsum", lib.ptr(mag, f0 * nb), fr, nb, nb, lib.ptr(sub), 0)
            lib.call("mstts_stft_fft", lib.ptr(cat), lib.ptr(one), lib.ptr(one), 1, 0.97, lib.ptr(hann), lib.ptr(tw), *common,
                     lib.ptr(mel, f0 * num_mels) if want_mel else None, lib.ptr(spec, f0 * nb) if want_spec else None, fr,
                     lib.ptr(mag, f0 * nb), lib.ptr(sub), 0.1 / fr, flags)
            f0 += fr
    out, f0 = [], 0
    for fr in frames:
        out.append((mel[f0:f0 + fr] if want_mel else None, spec[f0:f0 + fr] if want_spec else None))
        f0 += fr
    return out


def _fft_ok(num_freq, frame_shift_ms, frame_length_ms, sample_rate):
    n_fft, hop, win = _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    return bool(lib.load().mstts_stft_fft_supported(n_fft, win))


def spectrogram(y, num_freq, frame_shift_ms, frame_length_ms, sample_rate, ref_level_db=20, spectral_subtract=False, device="cuda",
                return_tensor=False):
    """Audio.py:19-22: normalised linear spectrogram [num_freq, frames] in [0, 1]."""
    (_, spec), = stft_features([y], num_freq, frame_shift_ms, frame_length_ms, sample_rate, ref_level_db=ref_level_db, want_mel=False,
                               want_spec=True, spectral_subtract=spectral_subtract, device=device)
    spec = spec.t()
    return spec if return_tensor else spec.cpu().numpy()


def spectrogram_and_mel(y, num_freq, frame_shift_ms, frame_length_ms, sample_rate, spect_ref_level_db=20, num_mels=80, max_abs_mels=None,
                        spectral_subtract=False, device="cuda", return_tensor=False):
    """Audio.py:34-40: both features from one STFT -> (spectrogram [num_freq, frames], mel [num_mels, frames])."""
    (mel, spec), = stft_features([y], num_freq, frame_shift_ms, frame_length_ms, sample_rate, num_mels=num_mels, max_abs_value=max_abs_mels,
                                 ref_level_db=spect_ref_level_db, want_mel=True, want_spec=True, spectral_subtract=spectral_subtract,
                                 device=device)
    spec, mel = spec.t(), mel.t()
    return (spec, mel) if return_tensor else (spec.cpu().numpy(), mel.cpu().numpy())


def melspectrogram(y, num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, max_abs_value=None,
                   spectral_subtract=False, device="cuda", return_tensor=False, use_fft=True):
    """Same signature and result layout ([num_mels, frames]) as the reference function.  One launch (FFT in LDS) when n_fft is a
    power of two - the reference's 2048 is; the DFT-as-GEMM form below covers any other size (use_fft=False forces it)."""
    if (use_fft or spectral_subtract or max_abs_value is None) and _fft_ok(num_freq, frame_shift_ms, frame_length_ms, sample_rate):
        (mel, _), = stft_features([y], num_freq, frame_shift_ms, frame_length_ms, sample_rate, num_mels=num_mels, max_abs_value=max_abs_value,
                                  spectral_subtract=spectral_subtract, device=device)
        mel = mel.t()
        return mel if return_tensor else mel.cpu().numpy()
    if spectral_subtract or max_abs_value is None:
        raise ValueError("spectral_subtract / the [0, 1] normalisation need a power-of-two transform size (the one-launch FFT path)")
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


# ---------------------------------------------------------------------------------------------------------------------
# Spectrogram -> waveform (Griffin-Lim), the export leg of Tacotron2.Inference with the Taco1 vocoder
# (Audio.py:15-27,50-60,84-99; Taco1_Mel_to_Spect/Modules.py:110-119; MSTTS_SV.py:403-412).  Host plumbing like in
# the reference (it runs in the export thread there; BASELINE config 1 calls it "plumbing, no GPU"): NumPy rFFTs.
# ---------------------------------------------------------------------------------------------------------------------
def _padded_window(n_fft, win_length):
    """librosa.util.pad_center(scipy.signal.get_window('hann', win_length, fftbins=True), n_fft)."""
    w = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win_length) / win_length)
    return w


def _stft(y, num_freq, frame_shift_ms, frame_length_ms, sample_rate):
    """librosa.stft(y, n_fft, hop_length, win_length): center=True, reflect padding -> [num_freq, 1 + len(y) // hop]."""
    n_fft, hop, win = _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    w = _padded_window(n_fft, win)
    yp = np.pad(np.asarray(y, np.float64), n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(yp[idx] * w[None, :], axis=1).T


def _istft(D, num_freq, frame_shift_ms, frame_length_ms, sample_rate):
    """librosa.istft(D, hop_length, win_length): windowed overlap-add of the inverse rFFTs, divided by the summed squared
    window where that is non-negligible, centre padding (n_fft // 2 each side) removed."""
    n_fft, hop, win = _stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    w = _padded_window(n_fft, win)
    n_frames = D.shape[1]
    frames = np.fft.irfft(D.T, n=n_fft, axis=1) * w[None, :]
    n = n_fft + hop * (n_frames - 1)
    y, wss = np.zeros(n), np.zeros(n)
    for i in range(n_frames):
        y[i * hop:i * hop + n_fft] += frames[i]
        wss[i * hop:i * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:n - n_fft // 2]


def inv_preemphasis(x, preemphasis=0.97):
    """scipy.signal.lfilter([1], [1, -preemphasis], x) (Audio.py:15-16)."""
    from scipy import signal
    return signal.lfilter([1], [1, -preemphasis], x)


def _denormalize(S, min_level_db=-100):
    return (np.clip(S, 0, 1) * -min_level_db) + min_level_db


def _db_to_amp(x):
    return np.power(10.0, x * 0.05)


def _griffin_lim(S, num_freq, frame_shift_ms, frame_length_ms, sample_rate, griffin_lim_iters=60, rng=None):
    """Audio.py:50-60: random initial phase, then `iters` rounds of istft -> stft -> keep the phase."""
    rng = rng if rng is not None else np.random
    angles = np.exp(2j * np.pi * rng.rand(*S.shape))
    S_complex = np.abs(S).astype(np.complex128)
    args = (num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    y = _istft(S_complex * angles, *args)
    for _ in range(griffin_lim_iters):
        angles = np.exp(1j * np.angle(_stft(y, *args)))
        y = _istft(S_complex * angles, *args)
    return y


def inv_spectrogram(spectrogram, num_freq, frame_shift_ms, frame_length_ms, sample_rate, ref_level_db=20, power=1.5,
                    griffin_lim_iters=60, rng=None):
    """Same signature as the reference (Audio.py:24-27); spectrogram is [num_freq, frames], normalised to [0, 1]."""
    S = _db_to_amp(_denormalize(np.asarray(spectrogram, np.float64)) + ref_level_db)
    return inv_preemphasis(_griffin_lim(S ** power, num_freq, frame_shift_ms, frame_length_ms, sample_rate,
                                        griffin_lim_iters=griffin_lim_iters, rng=rng))


def Griffin_Lim(spectrogram, rng=None):
    """Taco1_Mel_to_Spect/Modules.py:110-119: spectrogram [Time, Dim] -> waveform at hp.Sound.Sample_Rate."""
    from . import Hyper_Parameters as hp
    return inv_spectrogram(np.asarray(spectrogram).transpose(), num_freq=hp.Sound.Spectrogram_Dim, frame_shift_ms=hp.Sound.Frame_Shift,
                           frame_length_ms=hp.Sound.Frame_Length, sample_rate=hp.Sound.Sample_Rate,
                           griffin_lim_iters=hp.Taco1_Mel_to_Spect.Griffin_Lim_Iteration, rng=rng)
