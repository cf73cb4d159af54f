"""Independent NumPy-fp64 restatements of the core cells (TEST INFRASTRUCTURE).  They are written
from the reference source separately from oracle/model.py (torch) so the two can cross-check each
other - the only pin available, since the TF1 reference cannot run here (see oracle/__init__).
"""
import numpy as np


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def zoneout_lstm_cell(x, c_prev, h_prev, kernel, bias, zc=None, zh=None, rate=0.1):
    """ZoneoutLSTMCell.py:228-264: gates i,j,f,o; forget_bias 1.0; output = un-zoned m; state zoned."""
    g = np.concatenate([x, h_prev], 1) @ kernel + bias
    H = c_prev.shape[1]
    i, j, f, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c = sigmoid(f + 1.0) * c_prev + sigmoid(i) * np.tanh(j)
    m = sigmoid(o) * np.tanh(c)
    dc, dm = c - c_prev, m - h_prev
    if zc is not None:
        dc, dm = dc * zc, dm * zh
    return m, (1 - rate) * dc + c_prev, (1 - rate) * dm + h_prev


def conv1d_same(x, kernel, bias=None):
    """tf.layers.conv1d 'same' on [B,T,Cin], kernel [K,Cin,Cout]: explicit loops over taps."""
    B, T, _ = x.shape
    K, _, cout = kernel.shape
    left = (K - 1) // 2
    y = np.zeros((B, T, cout))
    for k in range(K):
        lo, hi = max(0, left - k), min(T, T + left - k)
        if hi > lo:
            y[:, lo:hi] += x[:, lo + k - left:hi + k - left] @ kernel[k]
    return y if bias is None else y + bias


def lsa_step(keys, values, lengths, query, cum, wq, conv_k, conv_b, dense_k, w, b):
    """Location_Sensitive_Attention.py:43-85 + masked softmax + context (quirks Q4-Q6)."""
    q = query @ wq
    f = conv1d_same(cum[:, :, None], conv_k, conv_b)
    loc = f @ dense_k
    e = (w.reshape(-1) * np.tanh(keys + q[:, None, :] + loc + b.reshape(-1))).sum(2)
    mask = np.arange(keys.shape[1])[None, :] < np.asarray(lengths)[:, None]
    e = np.where(mask, e, -np.inf)
    e = e - e.max(1, keepdims=True)
    a = np.exp(e)
    a /= a.sum(1, keepdims=True)
    return a, cum + a, (a[:, :, None] * values).sum(1)


def batch_norm_train(x, gamma, beta, eps=1e-3):
    mean = x.reshape(-1, x.shape[-1]).mean(0)
    var = ((x - mean) ** 2).reshape(-1, x.shape[-1]).mean(0)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var
