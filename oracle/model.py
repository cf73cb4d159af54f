"""torch-CPU restatement of the reference's Tacotron2 graph (TEST INFRASTRUCTURE - see __init__).

Every function cites the reference file:line it restates.  All randomness (dropout, zoneout)
enters through explicit 0/1 keep-masks so the HIP path can be compared on identical draws.
dtype is a parameter: float64 gives the golden values, float32 is the timed CPU baseline.
Gradients come from torch autograd over this forward; the optimizer is restated in train.py.

Layouts are the reference's: activations [B, T, C], conv kernels [K, Cin, Cout], dense kernels
[in, out], LSTM kernels [in + H, 4H] with gate order i, j, f, o.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import math
import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    """Model widths; defaults are Hyper_Parameters.py:4-62,93-163."""
    n_tok: int = 42
    emb: int = 512
    enc_conv_n: int = 3
    enc_conv_k: int = 5
    enc_conv_ch: int = 512
    enc_lstm: int = 256
    spk: int = 256
    att: int = 128
    att_k: int = 31
    att_ch: int = 32
    prenet_n: int = 2
    prenet: int = 256
    dec_lstm: int = 1024
    dec_lstm_n: int = 2
    n_mel: int = 80
    post_n: int = 5
    post_k: int = 5
    post_ch: int = 512
    max_inf: int = 1000
    zoneout: float = 0.1
    conv_drop: float = 0.5
    prenet_drop: float = 0.5
    # Taco1 mel->spectrogram
    bank_k: int = 8
    bank_ch: int = 128
    proj1_ch: int = 256
    proj1_k: int = 3
    proj2_k: int = 3
    highway_n: int = 4
    birnn: int = 128
    n_spec: int = 1025
    # speaker encoder
    spk_lstm: int = 256
    spk_lstm_n: int = 3
    spk_samples: int = 5
    spk_frames: int = 64

    @property
    def mem(self):
        return 2 * self.enc_lstm + self.spk


BN_EPS = 1e-3      # tf.layers.batch_normalization default epsilon
BN_MOM = 0.99      # tf.layers.batch_normalization default momentum


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------
def _glorot(rng, shape):
    """TF glorot_uniform (default initializer of tf.get_variable / tf.layers)."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


P_LSA = "decoder/decoder/attention_wrapper/location_sensitive_attention/"
P_CELL = "decoder/decoder/attention_wrapper/multi_rnn_cell/cell_%d/zoneout_lstm_cell/"
P_ENC_CELL = "encoder/bilstm/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/"
P_V = "mel_to_spectrogram/"
P_S = "speaker_embedding/"


def param_specs(d: Dims):
    """Ordered [(name, shape, init)] for every variable; names are the inferred TF variable names
    (SURVEY.md section 7).  init in {glorot, zeros, ones, const:<v>}."""
    s = []

    def bn(prefix, ch):
        s.extend([(prefix + "gamma", (ch,), "ones"), (prefix + "beta", (ch,), "zeros"),
                  (prefix + "moving_mean", (ch,), "zeros"), (prefix + "moving_variance", (ch,), "ones")])

    # encoder (Modules.py:15-73)
    s.append(("encoder/embedding_variable", (d.n_tok, d.emb), "glorot"))
    cin = d.emb
    for i in range(d.enc_conv_n):
        s.append(("encoder/conv_%d/conv1d/kernel" % i, (d.enc_conv_k, cin, d.enc_conv_ch), "glorot"))
        s.append(("encoder/conv_%d/conv1d/bias" % i, (d.enc_conv_ch,), "zeros"))
        bn("encoder/conv_%d/batch_normalization/" % i, d.enc_conv_ch)
        cin = d.enc_conv_ch
    for dr in ("fw", "bw"):
        s.append((P_ENC_CELL % dr + "kernel", (cin + d.enc_lstm, 4 * d.enc_lstm), "glorot"))
        s.append((P_ENC_CELL % dr + "bias", (4 * d.enc_lstm,), "zeros"))
    # attention memory layer (Location_Sensitive_Attention.py:36-41 -> BahdanauAttention)
    s.append(("attention/memory_layer/kernel", (d.mem, d.att), "glorot"))
    # decoder prenet (Modules.py:239-255)
    cin = d.n_mel
    for i in range(d.prenet_n):
        s.append(("decoder/decoder/prenet_%d/dense/kernel" % i, (cin, d.prenet), "glorot"))
        s.append(("decoder/decoder/prenet_%d/dense/bias" % i, (d.prenet,), "zeros"))
        cin = d.prenet
    # decoder LSTM stack (Modules.py:80-88); cell 0 input = prenet + 2*mem (quirk Q1)
    cin = d.prenet + 2 * d.mem
    for i in range(d.dec_lstm_n):
        s.append((P_CELL % i + "kernel", (cin + d.dec_lstm, 4 * d.dec_lstm), "glorot"))
        s.append((P_CELL % i + "bias", (4 * d.dec_lstm,), "zeros"))
        cin = d.dec_lstm
    # location sensitive attention (Location_Sensitive_Attention.py:43-85)
    s.append((P_LSA + "query_layer/kernel", (d.dec_lstm, d.att), "glorot"))
    s.append((P_LSA + "attention_convolution_dense_layer/conv1d/kernel", (d.att_k, 1, d.att_ch), "glorot"))
    s.append((P_LSA + "attention_convolution_dense_layer/conv1d/bias", (d.att_ch,), "zeros"))
    s.append((P_LSA + "attention_convolution_dense_layer/dense/kernel", (d.att_ch, d.att), "glorot"))
    s.append((P_LSA + "score_layer/weight_w", (1, 1, d.att), "glorot"))
    s.append((P_LSA + "score_layer/bias_b", (1, 1, d.att), "zeros"))
    # projection (Modules.py:309-321)
    s.append(("decoder/decoder/linear_projection/dense/kernel", (d.dec_lstm + d.mem, d.n_mel + 1), "glorot"))
    s.append(("decoder/decoder/linear_projection/dense/bias", (d.n_mel + 1,), "zeros"))
    # postnet (Modules.py:121-143)
    cin = d.n_mel
    for i in range(d.post_n):
        cout = d.post_ch if i < d.post_n - 1 else d.n_mel
        s.append(("decoder/conv_%d/conv1d/kernel" % i, (d.post_k, cin, cout), "glorot"))
        s.append(("decoder/conv_%d/conv1d/bias" % i, (cout,), "zeros"))
        bn("decoder/conv_%d/batch_normalization/" % i, cout)
        cin = cout
    # Taco1 mel->spectrogram (Taco1_Mel_to_Spect/Modules.py:8-105)
    for k in range(1, d.bank_k + 1):
        sfx = "" if k == 1 else "_%d" % (k - 1)
        s.append((P_V + "convbank_0/conv1d%s/kernel" % sfx, (k, d.n_mel, d.bank_ch), "glorot"))
        s.append((P_V + "convbank_0/conv1d%s/bias" % sfx, (d.bank_ch,), "zeros"))
        bn(P_V + "convbank_0/batch_normalization%s/" % sfx, d.bank_ch)
    s.append((P_V + "convbank_0/conv1d_8/kernel", (d.proj1_k, d.bank_k * d.bank_ch, d.proj1_ch), "glorot"))
    s.append((P_V + "convbank_0/conv1d_8/bias", (d.proj1_ch,), "zeros"))
    bn(P_V + "convbank_0/batch_normalization_8/", d.proj1_ch)
    s.append((P_V + "convbank_0/conv1d_9/kernel", (d.proj2_k, d.proj1_ch, d.n_mel), "glorot"))
    s.append((P_V + "convbank_0/conv1d_9/bias", (d.n_mel,), "zeros"))
    bn(P_V + "convbank_0/batch_normalization_9/", d.n_mel)
    for i in range(d.highway_n):
        s.append((P_V + "highway_%d/dense/kernel" % i, (d.n_mel, d.n_mel), "glorot"))
        s.append((P_V + "highway_%d/dense/bias" % i, (d.n_mel,), "zeros"))
        s.append((P_V + "highway_%d/dense_1/kernel" % i, (d.n_mel, d.n_mel), "glorot"))
        s.append((P_V + "highway_%d/dense_1/bias" % i, (d.n_mel,), "const:-1.0"))
    for dr in ("fw", "bw"):
        pre = P_V + "birnn/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/" % dr
        s.append((pre + "kernel", (d.n_mel + d.birnn, 4 * d.birnn), "glorot"))
        s.append((pre + "bias", (4 * d.birnn,), "zeros"))
    s.append((P_V + "dense/kernel", (2 * d.birnn, d.n_spec), "glorot"))
    s.append((P_V + "dense/bias", (d.n_spec,), "zeros"))
    # speaker encoder (Speaker_Embedding/Modules.py:6-37)
    s.append((P_S + "dense/kernel", (d.n_mel, d.spk), "glorot"))
    s.append((P_S + "dense/bias", (d.spk,), "zeros"))
    for i in range(d.spk_lstm_n):
        pre = P_S + "lstm/rnn/multi_rnn_cell/cell_%d/lstmcell_%d/" % (i, i)
        s.append((pre + "kernel", (d.spk + d.spk_lstm, 4 * d.spk_lstm), "glorot"))
        s.append((pre + "bias", (4 * d.spk_lstm,), "zeros"))
    return s


FROZEN_SCOPES = ("speaker_embedding", "mel_to_spectrogram", "waveglow")


def is_trainable(name):
    """MSTTS_SV.py:183-190: tacotron variables only; BN moving stats are not trainable."""
    if name.startswith(FROZEN_SCOPES):
        return False
    return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def in_weight_reg(name):
    """MSTTS_SV.py:145-159 membership by substring of the (lower-cased) variable name."""
    if not is_trainable(name):
        return False
    low = name.lower()
    return not any(t in low for t in ("bias", "embedding", "lstm", "rnn", "weight_w", "projection"))


def init_params(d: Dims, seed=1234):
    """numpy float64 dict, deterministic in (dims, seed)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, init in param_specs(d):
        if init == "glorot":
            out[name] = _glorot(rng, shape)
        elif init == "zeros":
            out[name] = np.zeros(shape)
        elif init == "ones":
            out[name] = np.ones(shape)
        else:
            out[name] = np.full(shape, float(init.split(":")[1]))
    return out


def to_torch(params, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------------
# primitive layers
# --------------------------------------------------------------------------------------------
def conv1d_same(x, kernel, bias=None, engine_gemm=False):
    """tf.layers.conv1d(padding='same', strides=1) on [B,T,Cin] with kernel [K,Cin,Cout].
    SAME padding: left=(K-1)//2, right=K-1-left (quirk Q12).  engine_gemm: one of the convolutions the HIP train engine runs as an
    implicit-im2col GEMM (rounded operands under GEMM_BF16)."""
    K, cin, cout = kernel.shape
    left = (K - 1) // 2
    xp = F.pad(x, (0, 0, left, K - 1 - left))
    win = xp.unfold(1, K, 1)                       # [B,T,Cin,K]
    win = win.permute(0, 1, 3, 2).reshape(x.shape[0], x.shape[1], K * cin)
    y = gmm(win, kernel.reshape(K * cin, cout)) if engine_gemm else win @ kernel.reshape(K * cin, cout)
    return y if bias is None else y + bias


def batch_norm(x, p, prefix, training, stats_out=None):
    """tf.layers.batch_normalization on the last axis of [B,T,C] (Modules.py:37-40; quirk Q11):
    training -> batch moments over (B,T) with biased variance; moving stats updated with
    momentum .99 from the same (biased) moments; inference -> moving stats."""
    if training:
        mean = x.mean(dim=(0, 1))
        var = ((x - mean) ** 2).mean(dim=(0, 1))
        if stats_out is not None:
            stats_out[prefix + "moving_mean"] = (p[prefix + "moving_mean"] * BN_MOM + mean.detach() * (1 - BN_MOM))
            stats_out[prefix + "moving_variance"] = (p[prefix + "moving_variance"] * BN_MOM + var.detach() * (1 - BN_MOM))
    else:
        mean, var = p[prefix + "moving_mean"], p[prefix + "moving_variance"]
    return (x - mean) * torch.rsqrt(var + BN_EPS) * p[prefix + "gamma"] + p[prefix + "beta"]


def dropout(x, keep_mask, rate):
    """tf.layers.dropout(training=True): x * mask / (1-rate)."""
    return x * keep_mask.to(x.dtype) / (1.0 - rate)


# BASELINE config 3 emulation ("bf16 with fp32 master"): when set, the decoder's recurrent products (both cells on the folded
# cell-0 kernel, the attention query) are computed as bf(X) . bf(W) (round to nearest even) in the working precision, their data
# gradients as bf(dY) . bf(W)^T, and their weight gradients from the unrounded operands - exactly what the HIP bf16 path does.
RECURRENT_BF16 = False


def _bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _BF16MatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _bf(x) @ _bf(w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return _bf(g) @ _bf(w).t(), x.t() @ g


# Full config-3 emulation: additionally every dense / conv contraction the HIP engine issues through its GEMM entry point (conv fwd /
# data gradient / weight gradient, prenet, hoisted cell-0 input product, projection, memory layer, encoder LSTM input product, and
# ALL hoisted weight-gradient GEMMs incl. the recurrent kernels') multiplies bf16-rounded operands, fp32/fp64 accumulate.
GEMM_BF16 = False


class _BF16MatMulFull(torch.autograd.Function):
    """y = bf(x) . bf(w); dx = bf(g) . bf(w)^T; dw = bf(x)^T . bf(g)   (x may carry leading batch dims)."""
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _bf(x) @ _bf(w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        x2, g2 = x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1])
        return _bf(g) @ _bf(w).t(), _bf(x2).t() @ _bf(g2)


class _MatMulWgradBF16(torch.autograd.Function):
    """Recurrent product that stays fp32 in the loop (encoder BiLSTM h . Wh: exact forward and data gradient) but whose hoisted
    weight gradient runs on the bf16 GEMM: dw = bf(x)^T . bf(g)."""
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w.t(), _bf(x).t() @ _bf(g)


def gmm(x, w):
    """A contraction the HIP engine runs through its GEMM entry point."""
    return _BF16MatMulFull.apply(x, w) if GEMM_BF16 else x @ w


def rmm(x, w):
    """Recurrent matmul of the decoder loop (bf16 skinny kernels in config-3 mode; its hoisted weight gradient is a GEMM)."""
    if RECURRENT_BF16:
        return _BF16MatMulFull.apply(x, w) if GEMM_BF16 else _BF16MatMul.apply(x, w)
    return _MatMulWgradBF16.apply(x, w) if GEMM_BF16 else x @ w


def zoneout_lstm_cell(x, c_prev, h_prev, kernel, bias, zc, zh, rate, training, gates=None):
    """ZoneoutLSTMCell.call (ZoneoutLSTMCell.py:188-271; quirks Q2,Q3).
    zc/zh: 0/1 keep masks (used only when training).  Returns (m, c_state, h_state)."""
    if gates is None:
        gates = torch.cat([x, h_prev], dim=1) @ kernel + bias
    i, j, f, o = gates.chunk(4, dim=1)
    c = torch.sigmoid(f + 1.0) * c_prev + torch.sigmoid(i) * torch.tanh(j)
    m = torch.sigmoid(o) * torch.tanh(c)
    dc, dm = c - c_prev, m - h_prev
    if training:
        dc, dm = dc * zc.to(x.dtype), dm * zh.to(x.dtype)
    return m, (1.0 - rate) * dc + c_prev, (1.0 - rate) * dm + h_prev


def run_lstm(x, lengths, kernel, bias, H, zc, zh, rate, training, reverse=False, residual=False, engine_gemm=False):
    """tf.nn.dynamic_rnn over one ZoneoutLSTMCell on [B,T,Cin]: past ``lengths`` the output is
    zero and the state is carried through unchanged; ``reverse`` = tf.reverse_sequence by length
    before and after (bidirectional_dynamic_rnn backward direction).  zc/zh: [T,B,H] in
    *processing* order.  lengths None -> full length."""
    B, T, _ = x.shape
    c = x.new_zeros(B, H)
    h = x.new_zeros(B, H)
    if lengths is None:
        lengths = torch.full((B,), T, dtype=torch.long)
    lengths = lengths.long()
    ar = torch.arange(T)
    if reverse:
        idx = torch.where(ar[None, :] < lengths[:, None], lengths[:, None] - 1 - ar[None, :], ar[None, :])
        x = torch.gather(x, 1, idx[:, :, None].expand(-1, -1, x.shape[2]))
    outs = []
    for t in range(T):
        gates = None
        if engine_gemm and GEMM_BF16:        # the train engine hoists x . Wx into one GEMM (rounded operands); h . Wh stays fp32 in the loop
            cin = x.shape[2]
            gates = gmm(x[:, t], kernel[:cin]) + _MatMulWgradBF16.apply(h, kernel[cin:]) + bias
        m, c2, h2 = zoneout_lstm_cell(x[:, t], c, h, kernel, bias,
                                      None if zc is None else zc[t], None if zh is None else zh[t],
                                      rate, training, gates=gates)
        if residual:
            m = m + x[:, t]
        live = (t < lengths)[:, None]
        outs.append(torch.where(live, m, torch.zeros_like(m)))
        c = torch.where(live, c2, c)
        h = torch.where(live, h2, h)
    y = torch.stack(outs, dim=1)
    if reverse:
        y = torch.gather(y, 1, idx[:, :, None].expand(-1, -1, H))
    return y


KINK_BAND = 1e-4      # |pre-activation| below which two correct implementations may disagree on which side of the ReLU kink it lies
RELU_INJECTED = {"elements": 0, "differ": 0}      # running count of injected-pattern elements and of those that differ from this evaluation's own pattern


def relu_at(pre, active=None):
    """ReLU (Modules.py:29-35).  `active` (optional bool tensor of pre's shape) fixes which elements count as active, exactly
    like an injected dropout mask: with millions of pre-activations a few land within rounding distance of 0, where an fp32 and
    an fp64 evaluation legitimately fall on different sides of the kink and the GRADIENT jumps.  Parity tests pass the
    pattern of the implementation under test; it may differ from this evaluation's own pattern only inside +-KINK_BAND
    (asserted), so it cannot hide a wrong ReLU."""
    if active is None:
        return torch.relu(pre)
    active = torch.as_tensor(active).to(torch.bool).reshape(pre.shape)
    differ = active != (pre.detach() > 0)
    RELU_INJECTED["elements"] += int(differ.numel())
    RELU_INJECTED["differ"] += int(differ.sum())
    if bool(differ.any()):
        worst = float(pre.detach().abs()[differ].max())
        assert worst < KINK_BAND, "injected ReLU pattern differs outside the kink band: |pre| = %g" % worst
    return pre * active.to(pre.dtype)


# --------------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------------
def encoder(p, d: Dims, token, token_length, training, masks, stats_out=None):
    """Encoder_Embedding / Encoder_Conv / Encoder_BiLSTM (Modules.py:15-73).
    masks: 'enc_conv_drop_%d' [B,T,C]; 'enc_zc_fw','enc_zh_fw','enc_zc_bw','enc_zh_bw' [T,B,H]."""
    x = p["encoder/embedding_variable"][token.long()]
    for i in range(d.enc_conv_n):
        pre = "encoder/conv_%d/" % i
        x = relu_at(conv1d_same(x, p[pre + "conv1d/kernel"], p[pre + "conv1d/bias"], engine_gemm=True), masks.get("relu_enc_%d" % i) if masks else None)
        x = batch_norm(x, p, pre + "batch_normalization/", training, stats_out)
        if training:
            x = dropout(x, masks["enc_conv_drop_%d" % i], d.conv_drop)
    outs = []
    for dr in ("fw", "bw"):
        outs.append(run_lstm(x, token_length, p[P_ENC_CELL % dr + "kernel"], p[P_ENC_CELL % dr + "bias"],
                             d.enc_lstm, masks.get("enc_zc_" + dr), masks.get("enc_zh_" + dr),
                             d.zoneout, training, reverse=(dr == "bw"), engine_gemm=True))
    return torch.cat(outs, dim=2)


def prenet(p, d: Dims, x, masks, step):
    """Decoder_Helper.prenet (Modules.py:239-255): dropout ALWAYS on (quirk Q9).
    masks 'prenet_drop_%d' [S,B,P] indexed by decoder step."""
    for i in range(d.prenet_n):
        pre = "decoder/decoder/prenet_%d/dense/" % i
        x = torch.relu(gmm(x, p[pre + "kernel"]) + p[pre + "bias"])
        x = dropout(x, masks["prenet_drop_%d" % i][step], d.prenet_drop)
    return x


def lsa_step(p, d: Dims, keys, values, length_mask, query_in, cum):
    """Location_Sensitive_Attention.__call__/score (Location_Sensitive_Attention.py:43-85) plus
    BahdanauAttention's -inf score mask + softmax and AttentionWrapper's context (quirks Q4-Q6).
    keys [B,T,A], values [B,T,M], length_mask bool [B,T], query_in [B,H], cum [B,T]."""
    q = rmm(query_in, p[P_LSA + "query_layer/kernel"])                               # [B,A]
    f = conv1d_same(cum[:, :, None], p[P_LSA + "attention_convolution_dense_layer/conv1d/kernel"],
                    p[P_LSA + "attention_convolution_dense_layer/conv1d/bias"])      # [B,T,32]
    loc = f @ p[P_LSA + "attention_convolution_dense_layer/dense/kernel"]             # [B,T,A]
    w = p[P_LSA + "score_layer/weight_w"].reshape(-1)
    b = p[P_LSA + "score_layer/bias_b"].reshape(-1)
    energy = (w * torch.tanh(keys + q[:, None, :] + loc + b)).sum(dim=2)              # [B,T]
    energy = torch.where(length_mask, energy, torch.full_like(energy, -float("inf")))
    align = torch.softmax(energy, dim=1)
    ctx = (align[:, :, None] * values).sum(dim=1)
    return align, cum + align, ctx


def attention_memory(p, memory, token_length):
    """_BaseAttentionMechanism._prepare_memory + memory_layer: values = memory zeroed past
    length, keys = values @ W_mem (no bias)."""
    B, T, _ = memory.shape
    mask = torch.arange(T)[None, :] < token_length.long()[:, None]
    values = memory * mask[:, :, None].to(memory.dtype)
    keys = gmm(values, p["attention/memory_layer/kernel"])
    return keys, values, mask


def decoder(p, d: Dims, memory, token_length, mel, mel_length, training, masks):
    """Decoder_LSTM + Decoder_Helper + Decoder_Decoder + Decoder_Dynamic_Decode
    (Modules.py:76-119,148-472) with TF AttentionWrapper semantics.

    training: teacher forcing, S = max(mel_length)+1 steps (quirk Q7).
    inference: free running until every row has emitted stop_logit >= 0 or time >= max_inf.
    masks: 'prenet_drop_%d' [S,B,P]; training only: 'dec_zc_%d','dec_zh_%d' [S,B,H].
    Returns dict(linear [B,S,n_mel], stop [B,S], align [B,T_enc,S])."""
    B = memory.shape[0]
    keys, values, lmask = attention_memory(p, memory, token_length)
    dt = memory.dtype
    c = [memory.new_zeros(B, d.dec_lstm) for _ in range(d.dec_lstm_n)]
    h = [memory.new_zeros(B, d.dec_lstm) for _ in range(d.dec_lstm_n)]
    ctx = memory.new_zeros(B, d.mem)
    cum = memory.new_zeros(B, memory.shape[1])
    frame = memory.new_zeros(B, d.n_mel)
    finished = torch.zeros(B, dtype=torch.bool)
    S_train = int(mel_length.max()) + 1 if training else None
    linear, stop, aligns = [], [], []
    t = 0
    while True:
        pre = prenet(p, d, frame, masks, t)
        x = torch.cat([pre, ctx, ctx], dim=1)            # quirk Q1: context enters twice
        for l in range(d.dec_lstm_n):
            gates = None
            if RECURRENT_BF16 or GEMM_BF16:      # the products as the HIP path forms them: the hoisted prenet part of cell 0 is a GEMM,
                K, bb = p[P_CELL % l + "kernel"], p[P_CELL % l + "bias"]     # [ctx | h0] runs on the FOLDED cell-0 kernel, [m0 | h1] on cell 1
                if l == 0:
                    Pn, Mm = d.prenet, d.mem
                    fold = torch.cat([K[Pn:Pn + Mm] + K[Pn + Mm:Pn + 2 * Mm], K[Pn + 2 * Mm:]], dim=0)
                    gates = gmm(pre, K[:Pn]) + bb + rmm(torch.cat([ctx, h[0]], dim=1), fold)
                else:
                    gates = rmm(torch.cat([x, h[l]], dim=1), K) + bb
            x, c[l], h[l] = zoneout_lstm_cell(
                x, c[l], h[l], p[P_CELL % l + "kernel"], p[P_CELL % l + "bias"],
                masks["dec_zc_%d" % l][t] if training else None,
                masks["dec_zh_%d" % l][t] if training else None, d.zoneout, training, gates=gates)
        align, cum, ctx = lsa_step(p, d, keys, values, lmask, x, cum)
        proj = gmm(torch.cat([x, ctx], dim=1), p["decoder/decoder/linear_projection/dense/kernel"]) \
            + p["decoder/decoder/linear_projection/dense/bias"]
        lin, st = proj[:, :d.n_mel], proj[:, d.n_mel]
        linear.append(lin); stop.append(st); aligns.append(align)
        if training:
            nf = t >= mel_length.long()
            frame = torch.zeros_like(frame) if bool(nf.all()) else mel[:, t]
        else:
            nf = (st >= 0.0) | torch.tensor(t >= d.max_inf)
            frame = lin
        finished = finished | nf
        t += 1
        if bool(finished.all()):
            break
    return {"linear": torch.stack(linear, 1), "stop": torch.stack(stop, 1), "align": torch.stack(aligns, 2)}


def postnet(p, d: Dims, x, training, masks, stats_out=None):
    """Decoder_Conv (Modules.py:121-143; quirk Q10): conv -> tanh -> BN -> dropout, all 5 layers."""
    for i in range(d.post_n):
        pre = "decoder/conv_%d/" % i
        x = torch.tanh(conv1d_same(x, p[pre + "conv1d/kernel"], p[pre + "conv1d/bias"], engine_gemm=True))
        x = batch_norm(x, p, pre + "batch_normalization/", training, stats_out)
        if training:
            x = dropout(x, masks["post_drop_%d" % i], d.conv_drop)
    return x


# --------------------------------------------------------------------------------------------
# Taco1 mel -> spectrogram, speaker encoder
# --------------------------------------------------------------------------------------------
def taco1_convbank(p, d: Dims, x, training, stats_out=None):
    """ConvBank (Taco1_Mel_to_Spect/Modules.py:8-52; quirks Q12,Q13)."""
    ys = []
    for k in range(1, d.bank_k + 1):
        sfx = "" if k == 1 else "_%d" % (k - 1)
        y = torch.relu(conv1d_same(x, p[P_V + "convbank_0/conv1d%s/kernel" % sfx], p[P_V + "convbank_0/conv1d%s/bias" % sfx]))
        ys.append(batch_norm(y, p, P_V + "convbank_0/batch_normalization%s/" % sfx, training, stats_out))
    y = torch.cat(ys, dim=2)
    # max_pooling1d(2, stride 1, 'same'): out[t] = max(y[t], y[t+1]), -inf pad on the right
    y = torch.maximum(y, torch.cat([y[:, 1:], torch.full_like(y[:, :1], -float("inf"))], dim=1))
    y = torch.relu(conv1d_same(y, p[P_V + "convbank_0/conv1d_8/kernel"], p[P_V + "convbank_0/conv1d_8/bias"]))
    y = batch_norm(y, p, P_V + "convbank_0/batch_normalization_8/", training, stats_out)
    y = conv1d_same(y, p[P_V + "convbank_0/conv1d_9/kernel"], p[P_V + "convbank_0/conv1d_9/bias"])
    y = batch_norm(y, p, P_V + "convbank_0/batch_normalization_9/", training, stats_out)
    return x + y


def taco1_highway(p, d: Dims, x):
    """Highway (Taco1_Mel_to_Spect/Modules.py:54-72)."""
    for i in range(d.highway_n):
        pre = P_V + "highway_%d/" % i
        Hh = torch.relu(x @ p[pre + "dense/kernel"] + p[pre + "dense/bias"])
        Tt = torch.sigmoid(x @ p[pre + "dense_1/kernel"] + p[pre + "dense_1/bias"])
        x = Hh * Tt + x * (1.0 - Tt)
    return x


def taco1_forward(p, d: Dims, mel, training, masks=None, stats_out=None):
    """ConvBank -> Highway -> BiRNN -> Projection as wired at MSTTS_SV.py:100-115."""
    masks = masks or {}
    x = taco1_highway(p, d, taco1_convbank(p, d, mel, training, stats_out))
    outs = []
    for dr in ("fw", "bw"):
        pre = P_V + "birnn/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/" % dr
        outs.append(run_lstm(x, None, p[pre + "kernel"], p[pre + "bias"], d.birnn,
                             masks.get("v_zc_" + dr), masks.get("v_zh_" + dr), d.zoneout, training,
                             reverse=(dr == "bw")))
    return torch.cat(outs, dim=2) @ p[P_V + "dense/kernel"] + p[P_V + "dense/bias"]


def speaker_encoder(p, d: Dims, spk_mel, training=False, masks=None):
    """Restructure / Stack_LSTM / Inference (Speaker_Embedding/Modules.py:6-37,127-137; quirks
    Q14,Q15) as wired at MSTTS_SV.py:49-56.  spk_mel [5B,64,80] -> [B,spk]."""
    masks = masks or {}
    x = spk_mel @ p[P_S + "dense/kernel"] + p[P_S + "dense/bias"]
    for i in range(d.spk_lstm_n):
        pre = P_S + "lstm/rnn/multi_rnn_cell/cell_%d/lstmcell_%d/" % (i, i)
        x = run_lstm(x, None, p[pre + "kernel"], p[pre + "bias"], d.spk_lstm,
                     masks.get("s_zc_%d" % i), masks.get("s_zh_%d" % i), d.zoneout, training,
                     residual=(i < d.spk_lstm_n - 1))
    e = x[:, -1, :].reshape(-1, d.spk_samples, d.spk).mean(dim=1)
    # tf.nn.l2_normalize(x) with axis=None: whole-tensor norm, epsilon 1e-12
    return e * torch.rsqrt(torch.clamp((e * e).sum(), min=1e-12))


# --------------------------------------------------------------------------------------------
# whole graph
# --------------------------------------------------------------------------------------------
def forward(p, d: Dims, batch, training, masks, stats_out=None, with_vocoder=True):
    """Tacotron2.Tensor_Generate forward (MSTTS_SV.py:45-125).
    batch: Token int [B,T], Token_Length [B], Mel [B,L,80], Mel_Length [B], and either
    'Speaker_Embedding' [B,spk] (config 2: random embeddings) or 'Speaker_Embedding_Mel' [5B,64,80]."""
    if "Speaker_Embedding" in batch:
        spk = batch["Speaker_Embedding"]
    else:
        spk = speaker_encoder(p, d, batch["Speaker_Embedding_Mel"], training, masks)
    enc = encoder(p, d, batch["Token"], batch["Token_Length"], training, masks, stats_out)
    memory = torch.cat([enc, spk[:, None, :].expand(-1, enc.shape[1], -1)], dim=2)
    dec = decoder(p, d, memory, batch["Token_Length"], batch["Mel"], batch["Mel_Length"], training, masks)
    post = postnet(p, d, dec["linear"], training, masks, stats_out)
    out = {"Linear": dec["linear"], "Mel": dec["linear"] + post, "Stop_Logit": dec["stop"],
           "Stop": torch.sigmoid(dec["stop"]), "Attention_History": dec["align"], "Memory": memory}
    if with_vocoder:
        out["Spectrogram"] = taco1_forward(p, d, out["Mel"], training, masks, stats_out)
    return out
