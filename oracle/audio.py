"""NumPy fp64 restatement of Audio.melspectrogram (TEST INFRASTRUCTURE; parity unpinned - librosa
is not installed, so librosa.stft / librosa.filters.mel are restated from their published
algorithms: librosa 0.6/0.7 `stft(center=True, pad_mode='reflect', window='hann')` and
`filters.mel(htk=False, norm=1)` i.e. Slaney scale with Slaney area normalisation).

Follows Audio.py:12-13 (preemphasis), :29-32 (melspectrogram), :42-48 (_magnitude), :62-64 (_stft),
:70-74 (_stft_parameters), :77-84 (mel basis), :87-88 (_amp_to_db), :95-96 (_symmetric_normalize).
"""
import numpy as np


def preemphasis(x, coef=0.97):
    """scipy.signal.lfilter([1, -coef], [1], x) (Audio.py:12-13)."""
    x = np.asarray(x, np.float64)
    y = x.copy()
    y[1:] -= coef * x[:-1]
    return y


def stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate):
    """Audio.py:70-74."""
    return (num_freq - 1) * 2, int(frame_shift_ms / 1000 * sample_rate), int(frame_length_ms / 1000 * sample_rate)


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)


def padded_window(n_fft, win_length):
    """librosa.util.pad_center(get_window('hann', win_length), n_fft)."""
    w = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def stft(y, n_fft, hop, win_length):
    """librosa.stft(center=True, pad_mode='reflect') -> [1+n_fft/2, 1+len(y)//hop] complex."""
    y = np.asarray(y, np.float64)
    w = padded_window(n_fft, win_length)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(yp[idx] * w[None, :], axis=1).T


def hz_to_mel(f):
    """Slaney (htk=False) scale."""
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels=80, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels) with defaults htk=False, norm=1 (Audio.py:82-84)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def magnitude(y, n_fft, hop, win, spectral_subtract=False):
    """Audio.py:42-48: |STFT| of the pre-emphasised signal; spectral_subtract removes a tenth of each bin's time mean, clipped at 0."""
    M = np.abs(stft(preemphasis(y), n_fft, hop, win))
    if spectral_subtract:
        M = np.clip(M - np.mean(M, axis=1, keepdims=True) / 10, a_min=0.0, a_max=np.inf)
    return M


def melspectrogram(y, num_freq=1025, frame_shift_ms=12.5, frame_length_ms=50, num_mels=80, sample_rate=16000, max_abs_value=4,
                   spectral_subtract=False):
    """Audio.py:29-32 -> [num_mels, frames]."""
    n_fft, hop, win = stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    M = magnitude(y, n_fft, hop, win, spectral_subtract)
    S = 20 * np.log10(np.maximum(1e-5, mel_basis(sample_rate, n_fft, num_mels) @ M))
    if max_abs_value is None:
        return np.clip((S + 100) / 100, 0, 1)
    return np.clip((2 * max_abs_value) * ((S + 100) / 100) - max_abs_value, -max_abs_value, max_abs_value)


def spectrogram(y, num_freq=1025, frame_shift_ms=12.5, frame_length_ms=50, sample_rate=16000, ref_level_db=20, spectral_subtract=False):
    """Audio.py:19-22 (+ :42-48, :88-92) -> normalised linear spectrogram [num_freq, frames] in [0, 1]."""
    n_fft, hop, win = stft_parameters(num_freq, frame_shift_ms, frame_length_ms, sample_rate)
    M = magnitude(y, n_fft, hop, win, spectral_subtract)
    S = 20 * np.log10(np.maximum(1e-5, M)) - ref_level_db
    return np.clip((S + 100) / 100, 0, 1)


def spectrogram_and_mel(y, num_freq=1025, frame_shift_ms=12.5, frame_length_ms=50, sample_rate=16000, spect_ref_level_db=20, num_mels=80,
                        max_abs_mels=4):
    """Audio.py:34-40: both features of one STFT."""
    return (spectrogram(y, num_freq, frame_shift_ms, frame_length_ms, sample_rate, spect_ref_level_db),
            melspectrogram(y, num_freq, frame_shift_ms, frame_length_ms, num_mels, sample_rate, max_abs_mels))
