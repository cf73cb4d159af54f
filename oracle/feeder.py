"""Integer/host contracts of the reference feeder (bit-exact work).

Follows Feeder.py:62-87 (speaker windows) and Feeder.py:186-206 (tokenisation + padding).
"""
import json
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DICT = os.path.join(_HERE, "..", "multi_speaker_tts_amd", "Token_Index_Dict.json")


def load_token_dict(path=_DICT):
    with open(path, "r") as f:
        return json.load(f)


def tokenize(text_list, token_dict=None):
    """Feeder.py:189-206: upper-case, <S> ... <E>, right-pad with <E>; unknown char -> KeyError."""
    d = token_dict or load_token_dict()
    rows = [np.array([d["<S>"]] + [d[ch] for ch in text.upper()] + [d["<E>"]], np.int32)
            for text in text_list]
    out = np.zeros((len(rows), max(r.shape[0] for r in rows)), np.int32) + d["<E>"]
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out, np.array([r.shape[0] for r in rows], np.int32)


def speaker_windows(mel_list, sample_nums=5, mel_frame=64, overlap=32, mel_dim=80):
    """Feeder.py:62-87.  mel_list: list of [T, mel_dim] -> float32 [len*sample_nums, mel_frame, mel_dim]."""
    required = sample_nums * (mel_frame - overlap) + overlap
    out = np.zeros((len(mel_list), sample_nums, mel_frame, mel_dim), np.float32)
    for i, mel in enumerate(mel_list):
        if mel.shape[0] < required:
            s = mel[:mel_frame]
            out[i, :, : s.shape[0]] = s
        else:
            for k in range(sample_nums):
                start = int((mel.shape[0] - required) / 2) + k * overlap
                out[i, k] = mel[start:start + mel_frame]
    return out.reshape(-1, mel_frame, mel_dim)


def window_starts(T, sample_nums=5, mel_frame=64, overlap=32):
    required = sample_nums * (mel_frame - overlap) + overlap
    if T < required:
        return None
    return [int((T - required) / 2) + k * overlap for k in range(sample_nums)]


def stop_cut(stop):
    """MSTTS_SV.py:395: first index with sigmoid(stop) > 0.5 (exclusive), else full length."""
    stop = np.asarray(stop)
    return int(np.argmax(stop > 0.5)) if np.any(stop > 0.5) else int(stop.shape[0])
