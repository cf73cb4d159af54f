"""Model dimensions, variable table and initialisation of the Tacotron2 graph.

Variable names follow the TF variable scopes of the reference graph (MSTTS_SV.py:45-125 builds
them under 'speaker_embedding', 'encoder', 'attention', 'decoder', 'mel_to_spectrogram'), shapes
and layouts are the reference's (conv [K,Cin,Cout], dense [in,out], LSTM [in+H,4H] i,j,f,o), so
a TF checkpoint's tensors map one-to-one.  All variables live in two flat fp32 device slabs:
the trainable tacotron variables (one contiguous range -> one fused Adam launch, one gradient
all-reduce) and everything else (BN moving statistics, frozen vocoder and speaker encoder).
"""
from __future__ import annotations

from dataclasses import dataclass
import math

import numpy as np
import torch

from . import Hyper_Parameters as hp


@dataclass
class Dims:
    n_tok: int = hp.Encoder.Embedding.Token_Size
    emb: int = hp.Encoder.Embedding.Embedding_Size
    enc_conv_n: int = hp.Encoder.Conv.Nums
    enc_conv_k: int = hp.Encoder.Conv.Kernel_Size
    enc_conv_ch: int = hp.Encoder.Conv.Channel
    enc_lstm: int = hp.Encoder.BiLSTM.Cell_Size
    spk: int = hp.Speaker_Embedding.Embedding_Size
    att: int = hp.Attention.Memory_Size
    att_k: int = hp.Attention.Conv.Kernel_Size
    att_ch: int = hp.Attention.Conv.Channel
    prenet_n: int = hp.Decoder.PreNet.Nums
    prenet: int = hp.Decoder.PreNet.Size
    dec_lstm: int = hp.Decoder.LSTM.Cell_Size
    dec_lstm_n: int = hp.Decoder.LSTM.Nums
    n_mel: int = hp.Sound.Mel_Dim
    post_n: int = hp.Decoder.Conv.Nums
    post_k: int = hp.Decoder.Conv.Kernel_Size
    post_ch: int = hp.Decoder.Conv.Channel
    max_inf: int = hp.Decoder.LSTM.Max_Inference_Length
    zoneout: float = hp.Decoder.LSTM.Zoneout_Rate
    conv_drop: float = hp.Encoder.Conv.Dropout_Rate
    prenet_drop: float = hp.Decoder.PreNet.Dropout_Rate
    bank_k: int = hp.Taco1_Mel_to_Spect.ConvBank.Max_Kernel_Size
    bank_ch: int = hp.Taco1_Mel_to_Spect.ConvBank.Channel
    proj1_ch: int = hp.Taco1_Mel_to_Spect.ConvBank.Projection1.Channel
    proj1_k: int = hp.Taco1_Mel_to_Spect.ConvBank.Projection1.Kernel_Size
    proj2_k: int = hp.Taco1_Mel_to_Spect.ConvBank.Projection2.Kernel_Size
    highway_n: int = hp.Taco1_Mel_to_Spect.Highway.Nums
    birnn: int = hp.Taco1_Mel_to_Spect.BiRNN.Cell_Size
    n_spec: int = hp.Sound.Spectrogram_Dim
    spk_lstm: int = hp.Speaker_Embedding.LSTM.Cell_Size
    spk_lstm_n: int = hp.Speaker_Embedding.LSTM.Nums
    spk_samples: int = hp.Speaker_Embedding.Inference.Sample_Nums
    spk_frames: int = hp.Speaker_Embedding.Inference.Mel_Frame

    @property
    def mem(self):
        return 2 * self.enc_lstm + self.spk


LSA = "decoder/decoder/attention_wrapper/location_sensitive_attention/"
CELL = "decoder/decoder/attention_wrapper/multi_rnn_cell/cell_%d/zoneout_lstm_cell/"
ENC_CELL = "encoder/bilstm/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/"
VOC = "mel_to_spectrogram/"
SPK = "speaker_embedding/"
VOC_CELL = VOC + "birnn/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/zoneout_lstm_cell/"
SPK_CELL = SPK + "lstm/rnn/multi_rnn_cell/cell_%d/lstmcell_%d/"
FROZEN = ("speaker_embedding", "mel_to_spectrogram", "waveglow")


def bank_suffix(k):
    """TF auto-numbering of the k-th (1-based) conv1d / batch_normalization in convbank_0."""
    return "" if k == 1 else "_%d" % (k - 1)


def variable_table(d: Dims):
    """[(name, shape, init)] in graph-construction order."""
    t = []

    def conv(prefix, k, cin, cout):
        t.append((prefix + "conv1d/kernel", (k, cin, cout), "glorot"))
        t.append((prefix + "conv1d/bias", (cout,), "zeros"))

    def bn(prefix, ch):
        t.extend([(prefix + "gamma", (ch,), "ones"), (prefix + "beta", (ch,), "zeros"),
                  (prefix + "moving_mean", (ch,), "zeros"), (prefix + "moving_variance", (ch,), "ones")])

    def cell(prefix, cin, H):
        t.append((prefix + "kernel", (cin + H, 4 * H), "glorot"))
        t.append((prefix + "bias", (4 * H,), "zeros"))

    def dense(prefix, cin, cout, bias="zeros"):
        t.append((prefix + "kernel", (cin, cout), "glorot"))
        if bias is not None:
            t.append((prefix + "bias", (cout,), bias))

    t.append(("encoder/embedding_variable", (d.n_tok, d.emb), "glorot"))
    cin = d.emb
    for i in range(d.enc_conv_n):
        conv("encoder/conv_%d/" % i, d.enc_conv_k, cin, d.enc_conv_ch)
        bn("encoder/conv_%d/batch_normalization/" % i, d.enc_conv_ch)
        cin = d.enc_conv_ch
    for dr in ("fw", "bw"):
        cell(ENC_CELL % dr, cin, d.enc_lstm)
    dense("attention/memory_layer/", d.mem, d.att, bias=None)
    cin = d.n_mel
    for i in range(d.prenet_n):
        dense("decoder/decoder/prenet_%d/dense/" % i, cin, d.prenet)
        cin = d.prenet
    cin = d.prenet + 2 * d.mem
    for i in range(d.dec_lstm_n):
        cell(CELL % i, cin, d.dec_lstm)
        cin = d.dec_lstm
    dense(LSA + "query_layer/", d.dec_lstm, d.att, bias=None)
    t.append((LSA + "attention_convolution_dense_layer/conv1d/kernel", (d.att_k, 1, d.att_ch), "glorot"))
    t.append((LSA + "attention_convolution_dense_layer/conv1d/bias", (d.att_ch,), "zeros"))
    dense(LSA + "attention_convolution_dense_layer/dense/", d.att_ch, d.att, bias=None)
    t.append((LSA + "score_layer/weight_w", (1, 1, d.att), "glorot"))
    t.append((LSA + "score_layer/bias_b", (1, 1, d.att), "zeros"))
    dense("decoder/decoder/linear_projection/dense/", d.dec_lstm + d.mem, d.n_mel + 1)
    cin = d.n_mel
    for i in range(d.post_n):
        cout = d.post_ch if i < d.post_n - 1 else d.n_mel
        conv("decoder/conv_%d/" % i, d.post_k, cin, cout)
        bn("decoder/conv_%d/batch_normalization/" % i, cout)
        cin = cout
    # frozen: Taco1 mel -> spectrogram
    for k in range(1, d.bank_k + 1):
        sfx = bank_suffix(k)
        t.append((VOC + "convbank_0/conv1d%s/kernel" % sfx, (k, d.n_mel, d.bank_ch), "glorot"))
        t.append((VOC + "convbank_0/conv1d%s/bias" % sfx, (d.bank_ch,), "zeros"))
        bn(VOC + "convbank_0/batch_normalization%s/" % sfx, d.bank_ch)
    t.append((VOC + "convbank_0/conv1d_8/kernel", (d.proj1_k, d.bank_k * d.bank_ch, d.proj1_ch), "glorot"))
    t.append((VOC + "convbank_0/conv1d_8/bias", (d.proj1_ch,), "zeros"))
    bn(VOC + "convbank_0/batch_normalization_8/", d.proj1_ch)
    t.append((VOC + "convbank_0/conv1d_9/kernel", (d.proj2_k, d.proj1_ch, d.n_mel), "glorot"))
    t.append((VOC + "convbank_0/conv1d_9/bias", (d.n_mel,), "zeros"))
    bn(VOC + "convbank_0/batch_normalization_9/", d.n_mel)
    for i in range(d.highway_n):
        dense(VOC + "highway_%d/dense/" % i, d.n_mel, d.n_mel)
        dense(VOC + "highway_%d/dense_1/" % i, d.n_mel, d.n_mel, bias="const:-1.0")
    for dr in ("fw", "bw"):
        cell(VOC_CELL % dr, d.n_mel, d.birnn)
    dense(VOC + "dense/", 2 * d.birnn, d.n_spec)
    # frozen: speaker encoder
    dense(SPK + "dense/", d.n_mel, d.spk)
    for i in range(d.spk_lstm_n):
        cell(SPK_CELL % (i, i), d.spk, d.spk_lstm)
    return t


def is_trainable(name):
    """MSTTS_SV.py:183-190."""
    return not name.startswith(FROZEN) and not name.endswith(("moving_mean", "moving_variance"))


def in_weight_reg(name):
    """MSTTS_SV.py:145-159: membership by substring of the lower-cased variable name."""
    low = name.lower()
    return is_trainable(name) and not any(s in low for s in ("bias", "embedding", "lstm", "rnn", "weight_w", "projection"))


def _glorot(rng, shape):
    if len(shape) == 1:
        fi = fo = shape[0]
    elif len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape)


def initial_values(d: Dims, seed=1234):
    """TF defaults: glorot-uniform kernels, zero biases, BN (1, 0, 0, 1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, init in variable_table(d):
        if init == "glorot":
            out[name] = _glorot(rng, shape).astype(np.float32)
        elif init == "zeros":
            out[name] = np.zeros(shape, np.float32)
        elif init == "ones":
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.full(shape, float(init.split(":")[1]), np.float32)
    return out


class ParamStore:
    """Two flat fp32 slabs on the device + named views.  Offsets are multiples of 4 floats so every
    variable starts 16-byte aligned (the kernels' float4 paths rely on it)."""

    def __init__(self, d: Dims, device, seed=1234, values=None, trainable_fn=None, weight_reg_fn=None):
        """trainable_fn / weight_reg_fn: name -> bool; default = the Tacotron2 trainer's sets (MSTTS_SV.py:145-159,183-190).
        The auxiliary trainers pass their own (e.g. the `mel_to_spectrogram` scope for the Taco1 vocoder trainer)."""
        self.dims = d
        self.table = variable_table(d)
        trainable_fn = trainable_fn or is_trainable
        weight_reg_fn = weight_reg_fn or in_weight_reg
        self.offset, self.shape, self.trainable = {}, {}, {}
        n_t = n_f = 0
        # the non-trainable slab starts with every BN moving statistic (one contiguous range `frozen[:n_moving]`: the train engine
        # snapshots it with one small copy before a speculative forward tail), the frozen sub-models' variables follow
        is_moving = lambda name: name.endswith(("moving_mean", "moving_variance"))
        for moving_pass in (True, False):
            for name, shape, _ in self.table:
                n = int(np.prod(shape))
                tr = bool(trainable_fn(name))
                if moving_pass != (not tr and is_moving(name)):
                    continue
                self.trainable[name] = tr
                self.shape[name] = tuple(shape)
                if tr:
                    self.offset[name] = n_t
                    n_t += (n + 3) // 4 * 4
                else:
                    self.offset[name] = n_f
                    n_f += (n + 3) // 4 * 4
            if moving_pass:
                self.n_moving = n_f
        self.n_train, self.n_frozen = n_t, n_f
        self.train = torch.zeros(n_t, dtype=torch.float32, device=device)
        self.frozen = torch.zeros(max(n_f, 4), dtype=torch.float32, device=device)
        self.grad = torch.zeros(n_t, dtype=torch.float32, device=device)
        self.adam_m = torch.zeros(n_t, dtype=torch.float32, device=device)
        self.adam_v = torch.zeros(n_t, dtype=torch.float32, device=device)
        wd = np.zeros(n_t, np.uint8)
        for name, shape, _ in self.table:
            if self.trainable[name] and weight_reg_fn(name):
                o = self.offset[name]
                wd[o:o + int(np.prod(shape))] = 1
        self.wd_mask = torch.from_numpy(wd).to(device)
        self.version = 0          # bumped whenever the variables change (load, optimizer step): engines key their packed / folded copies on it
        self.load(values if values is not None else initial_values(d, seed))

    def _slab(self, name, grad=False):
        if grad:
            return self.grad
        return self.train if self.trainable[name] else self.frozen

    def view(self, name, grad=False):
        o, shape = self.offset[name], self.shape[name]
        return self._slab(name, grad)[o:o + int(np.prod(shape))].view(shape)

    def p(self, name):
        """(slab tensor, element offset) - what the pointer plumbing wants."""
        return self._slab(name), self.offset[name]

    def g(self, name):
        return self.grad, self.offset[name]

    def touch(self):
        """The variables were written (optimizer step, broadcast, load): every cache keyed on `version` (packed / folded kernels of the
        inference engines, the train engine's derived copies) is stale from here on.  Every code path that writes `train` / `frozen` calls this."""
        self.version += 1

    def load(self, values):
        self.touch()
        for name, shape, _ in self.table:
            if name in values:
                v = values[name]
                v = torch.as_tensor(np.asarray(v, dtype=np.float32) if not torch.is_tensor(v) else v.detach().to(torch.float32).cpu().numpy())
                self.view(name).copy_(v.reshape(self.shape[name]))

    def export(self, grads=False):
        return {name: self.view(name, grads).detach().cpu().numpy().copy() for name, _, _ in self.table
                if (self.trainable[name] or not grads)}
