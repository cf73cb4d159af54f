"""Function-level mirror of the reference's ``Modules.py`` for callers that compose the graph
themselves: every function takes / returns device tensors and is a thin call into the engines (which
in turn only issue libmstts_hip.so calls).  Names and argument meaning follow Modules.py:15-143.
"""
from __future__ import annotations

import torch

from .inference import InferEngine


def _engine(engine=None):
    return engine if engine is not None else InferEngine()


def Encoder(token, token_length, speaker_embedding, engine=None):
    """Encoder_Embedding -> Encoder_Conv -> Encoder_BiLSTM -> speaker concat -> attention memory/keys
    (Modules.py:15-73, MSTTS_SV.py:58-83), inference mode.  Returns (values [B,T,768], keys [B,T,128])."""
    e = _engine(engine)
    return e.encoder(token.to(torch.int32).contiguous(), token_length.to(torch.int32).contiguous(), speaker_embedding.contiguous())


def Decoder_LSTM(values, keys, token_length, engine=None, masks=None, max_steps=None):
    """Decoder_LSTM in inference mode (Modules.py:76-119): step-major linear, stop logits, alignments, steps."""
    return _engine(engine).decode(values, keys, token_length, masks=masks, max_steps=max_steps)


def Decoder_Conv(linear, engine=None):
    """Decoder_Conv + residual (Modules.py:121-143, MSTTS_SV.py:93-97) on [B,S,80]."""
    B, S, _ = linear.shape
    return _engine(engine).postnet(linear.contiguous(), B, S)


def Mel_to_Spectrogram(mel, engine=None):
    """Taco1 ConvBank/Highway/BiRNN/Projection (Taco1_Mel_to_Spect/Modules.py:8-105)."""
    B, S, _ = mel.shape
    return _engine(engine).mel_to_spectrogram(mel.contiguous(), B, S)


def Speaker_Embedding(speaker_mel, engine=None):
    """Restructure/Stack_LSTM/Inference (Speaker_Embedding/Modules.py:6-37,127-137) on [5B,64,80]."""
    return _engine(engine).speaker_embedding(speaker_mel.contiguous())
