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


# ---------------------------------------------------------------------------------------------------------------------
# Second restatement of the whole Tacotron2 TRAIN forward, its loss and optimizer step, in plain NumPy fp64 with explicit
# loops - written from the reference sources (file:line cited per block), NOT from oracle/model.py, so that the wiring
# quirks have two independent statements: the doubled context (Q1), max(L)+1 decoder steps and the teacher-forcing shift
# (Q7), unmasked losses (Q8), weight-regularisation membership (Q19) and TF-Adam (Q18).  tests/test_cpu_oracle.py runs both
# on the same inputs and requires agreement to 1e-9.  No gradients here: the backward pass of model.py is pinned by finite
# differences instead.
# ---------------------------------------------------------------------------------------------------------------------
def _bn_dropout(x, p, prefix, keep_mask, rate, stats):
    """tf.layers.batch_normalization(training=True) on [B,T,C] then tf.layers.dropout (Modules.py:37-45,133-141): batch moments over
    (B,T) incl. padded frames, biased variance, eps 1e-3; moving <- .99 moving + .01 batch (variance biased too)."""
    y, mean, var = batch_norm_train(x, p[prefix + "gamma"], p[prefix + "beta"])
    stats[prefix + "moving_mean"] = 0.99 * p[prefix + "moving_mean"] + 0.01 * mean
    stats[prefix + "moving_variance"] = 0.99 * p[prefix + "moving_variance"] + 0.01 * var
    return y * keep_mask / (1.0 - rate)


def _dynamic_rnn(x, lengths, kernel, bias, H, zc, zh, rate, reverse):
    """tf.nn.dynamic_rnn / stack_bidirectional_dynamic_rnn over one ZoneoutLSTMCell (Modules.py:49-73): row b runs its own
    lengths[b] steps; the backward direction reads the sequence reversed BY LENGTH and its outputs are reversed back; past the
    length the output is zero and the state stays.  zc/zh are indexed by processing step."""
    B, T, _ = x.shape
    out = np.zeros((B, T, H))
    for b in range(B):
        c = np.zeros((1, H)); h = np.zeros((1, H))
        n = int(lengths[b])
        for step in range(n):
            pos = n - 1 - step if reverse else step
            m, c, h = zoneout_lstm_cell(x[b:b + 1, pos], c, h, kernel, bias, zc[step, b:b + 1], zh[step, b:b + 1], rate)
            out[b, pos] = m[0]
    return out


def tacotron_train_forward(p, d, batch, masks):
    """Tacotron2.Tensor_Generate with Is_Training = True (MSTTS_SV.py:45-98) on NumPy arrays.  p: name -> array; batch: Token,
    Token_Length, Mel, Mel_Length, Speaker_Embedding (config-2 style, the speaker encoder bypassed); masks: name -> 0/1 array.
    Returns Linear [B,S,n_mel], Mel, Stop_Logit [B,S], Attention_History [B,T,S] and the new BN moving statistics."""
    tok, tlen = np.asarray(batch["Token"]), np.asarray(batch["Token_Length"])
    mel, mlen = np.asarray(batch["Mel"], np.float64), np.asarray(batch["Mel_Length"])
    spk = np.asarray(batch["Speaker_Embedding"], np.float64)
    B, T = tok.shape
    stats = {}
    # --- encoder (Modules.py:15-73): embedding gather, 3 x (conv K5 SAME + bias -> ReLU -> BN -> dropout .5), BiLSTM
    x = p["encoder/embedding_variable"][tok]
    for i in range(d.enc_conv_n):
        pre = "encoder/conv_%d/" % i
        x = np.maximum(conv1d_same(x, p[pre + "conv1d/kernel"], p[pre + "conv1d/bias"]), 0.0)
        x = _bn_dropout(x, p, pre + "batch_normalization/", masks["enc_conv_drop_%d" % i], d.conv_drop, stats)
    enc = []
    for dr in ("fw", "bw"):
        cell = "encoder/bilstm/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/" % dr
        enc.append(_dynamic_rnn(x, tlen, p[cell + "kernel"], p[cell + "bias"], d.enc_lstm, masks["enc_zc_" + dr], masks["enc_zh_" + dr],
                                d.zoneout, dr == "bw"))
    # --- memory = [encoder | tiled speaker embedding] (MSTTS_SV.py:70-71); BahdanauAttention zeroes it past Token_Length and
    #     projects the keys with a bias-free dense (Location_Sensitive_Attention.py:36-41 -> _prepare_memory)
    memory = np.concatenate(enc + [np.repeat(spk[:, None, :], T, axis=1)], axis=2)
    live = np.arange(T)[None, :] < tlen[:, None]
    values = memory * live[:, :, None]
    keys = values @ p["attention/memory_layer/kernel"]
    # --- decoder loop (Modules.py:76-119,148-472 + AttentionWrapper): S = max(Mel_Length) + 1 steps, finished = time >= length
    S = int(mlen.max()) + 1
    H = d.dec_lstm
    lsa = "decoder/decoder/attention_wrapper/location_sensitive_attention/"
    c = [np.zeros((B, H)) for _ in range(d.dec_lstm_n)]
    h = [np.zeros((B, H)) for _ in range(d.dec_lstm_n)]
    ctx, cum = np.zeros((B, values.shape[2])), np.zeros((B, T))
    frame = np.zeros((B, d.n_mel))                                   # initial input: zeros (Modules.py:178-185)
    linear, stop, hist = [], [], []
    for t in range(S):
        y = frame
        for i in range(d.prenet_n):                                  # prenet: dropout always on (Modules.py:239-255)
            pre = "decoder/decoder/prenet_%d/dense/" % i
            y = np.maximum(y @ p[pre + "kernel"] + p[pre + "bias"], 0.0) * masks["prenet_drop_%d" % i][t] / (1.0 - d.prenet_drop)
        # helper concatenates [prenet, attention]; AttentionWrapper's default cell_input_fn concatenates attention AGAIN (Q1)
        y = np.concatenate([np.concatenate([y, ctx], 1), ctx], 1)
        for l in range(d.dec_lstm_n):
            cell = "decoder/decoder/attention_wrapper/multi_rnn_cell/cell_%d/zoneout_lstm_cell/" % l
            y, c[l], h[l] = zoneout_lstm_cell(y, c[l], h[l], p[cell + "kernel"], p[cell + "bias"], masks["dec_zc_%d" % l][t],
                                              masks["dec_zh_%d" % l][t], d.zoneout)
        a, cum, ctx = lsa_step(keys, values, tlen, y, cum, p[lsa + "query_layer/kernel"],
                               p[lsa + "attention_convolution_dense_layer/conv1d/kernel"], p[lsa + "attention_convolution_dense_layer/conv1d/bias"],
                               p[lsa + "attention_convolution_dense_layer/dense/kernel"], p[lsa + "score_layer/weight_w"], p[lsa + "score_layer/bias_b"])
        out = np.concatenate([y, ctx], 1) @ p["decoder/decoder/linear_projection/dense/kernel"] + p["decoder/decoder/linear_projection/dense/bias"]
        linear.append(out[:, :d.n_mel]); stop.append(out[:, d.n_mel]); hist.append(a)
        # next input (Modules.py:212-237): the target frame of THIS time index, zeros once every row is finished
        frame = np.zeros((B, d.n_mel)) if np.all(t >= mlen) else mel[:, t]
    lin = np.stack(linear, 1)
    # --- postnet (Modules.py:121-143): 5 x (conv K5 -> tanh -> BN -> dropout), residual (MSTTS_SV.py:93-97)
    x = lin
    for i in range(d.post_n):
        pre = "decoder/conv_%d/" % i
        x = np.tanh(conv1d_same(x, p[pre + "conv1d/kernel"], p[pre + "conv1d/bias"]))
        x = _bn_dropout(x, p, pre + "batch_normalization/", masks["post_drop_%d" % i], d.conv_drop, stats)
    return {"Linear": lin, "Mel": lin + x, "Stop_Logit": np.stack(stop, 1), "Attention_History": np.stack(hist, 2), "stats": stats}


def weight_regularised(name):
    """MSTTS_SV.py:145-159: trainable tacotron variables whose lower-cased name holds none of the listed substrings."""
    if name.split("/")[0] in ("speaker_embedding", "mel_to_spectrogram", "waveglow"):
        return False
    if name.endswith("moving_mean") or name.endswith("moving_variance"):
        return False
    low = name.lower()
    for word in ["bias", "embedding", "lstm", "rnn", "weight_w", "projection"]:
        if word in low:
            return False
    return True


def tacotron_losses(p, out, batch, wr_rate=1e-6, use_l1=True):
    """MSTTS_SV.py:127-161: stop target = NOT sequence_mask(Mel_Length, max + 1); MSE (+ L1) of the first max(L) output frames
    against the zero-padded targets as plain means over every element; mean sigmoid cross-entropy; 1e-6 * sum of tf.nn.l2_loss."""
    mel, mlen = np.asarray(batch["Mel"], np.float64), np.asarray(batch["Mel_Length"])
    S = int(mlen.max()) + 1
    target = np.zeros((mel.shape[0], S))
    for b in range(mel.shape[0]):
        target[b, int(mlen[b]):] = 1.0
    res = {}
    for key, name in (("Linear", "Linear_Loss"), ("Mel", "Postnet_Loss")):
        diff = out[key][:, :-1] - mel
        res[name] = np.mean(diff * diff) + (np.mean(np.abs(diff)) if use_l1 else 0.0)
    z = out["Stop_Logit"]
    res["Stop_Loss"] = np.mean(np.maximum(z, 0.0) - z * target + np.log(1.0 + np.exp(-np.abs(z))))
    res["Weight_Regularization_Loss"] = wr_rate * sum(0.5 * np.sum(np.square(v)) for k, v in p.items() if weight_regularised(k))
    res["Loss"] = res["Linear_Loss"] + res["Postnet_Loss"] + res["Stop_Loss"] + res["Weight_Regularization_Loss"]
    return res


def tf_learning_rate(step, initial=1e-3, minimum=1e-5, decay_step=10000, decay_rate=0.5):
    """MSTTS_SV.py:163-169: tf.train.exponential_decay (continuous exponent) clipped to [Min, Initial]."""
    return float(np.clip(initial * decay_rate ** (step / decay_step), minimum, initial))


def tf_adam(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, epsilon=1e-6):
    """tf.train.AdamOptimizer.apply (MSTTS_SV.py:171-176), `step` = global step BEFORE the update (t = step + 1):
    lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v updated first; theta -= lr_t m / (sqrt(v) + eps) - epsilon NOT bias-corrected."""
    t = step + 1
    m = m + (grad - m) * (1.0 - beta1)
    v = v + (grad * grad - v) * (1.0 - beta2)
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    return param - lr_t * m / (np.sqrt(v) + epsilon), m, v
