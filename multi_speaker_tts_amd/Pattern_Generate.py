"""The on-disk training-pattern format of the reference (Pattern_Generate.py:14-76,245-274): one pickle (protocol 2) per
utterance holding {'Token': int32[T_tok], 'Mel': float32[T_mel, 80], 'Text': str, 'Dataset': str}, named
'<DATASET>.<prefix><wav basename>.PICKLE', plus METADATA.PICKLE with the hyper parameters the set was made with and the
per-file lengths.  The dataset directory walkers (LJ / VCTK / TIMIT layouts, .sph decoding) are not rebuilt; this module is
what turns (wav, text) pairs into pattern files the feeder can train from.
"""
from __future__ import annotations

import os
import pickle
import re

import numpy as np

from . import Hyper_Parameters as hp
from . import Feeder as _Feeder

regex_Checker = re.compile("[A-Z\'\",.?!\\-&;:()\\[\\]\\s]+")


def Text_Filtering(text):
    """Pattern_Generate.py:14-31: upper-case, drop quotes / closing brackets, tidy spaces; None when the sentence holds
    characters outside the token set or starts with an apostrophe."""
    text = text.upper().strip()
    for ch in ['"', ")"]:
        text = text.replace(ch, "")
    for a, b in [(" ?", "?"), ("  ", " "), (" ,", ","), (" !", "!")]:
        text = text.replace(a, b)
    text = text.strip()
    found = regex_Checker.findall(text)
    if len(found) != 1 or text.startswith("'"):
        return None
    return found[0]


def Mel_Generate(path, range_Ignore=False, device="cuda"):
    """Pattern_Generate.py:33-58: load at hp.Sound.Sample_Rate, trim (top_db 15, librosa's default 2048/512 frames),
    scale by 0.99, reject by duration, mel through the HIP STFT kernel."""
    from . import Audio
    sig = _Feeder.load_wav(path, frame=2048, hop=512)
    ms = sig.shape[0] / hp.Sound.Sample_Rate * 1000
    if not range_Ignore and (ms < hp.Train.Use_Wav_Length_Range[0] or ms > hp.Train.Use_Wav_Length_Range[1]):
        return None
    return np.transpose(Audio.melspectrogram(y=sig, num_freq=hp.Sound.Spectrogram_Dim, frame_shift_ms=hp.Sound.Frame_Shift,
                                             frame_length_ms=hp.Sound.Frame_Length, num_mels=hp.Sound.Mel_Dim, sample_rate=hp.Sound.Sample_Rate,
                                             max_abs_value=hp.Sound.Max_Abs_Mel, device=device)).astype(np.float32)


def Pattern_File_Write(file_Name, text, mel, token_Index_Dict, dataset, pattern_path=None):
    """Pattern_Generate.py:66-76."""
    token = np.array([token_Index_Dict[letter] for letter in text]).astype(np.int32)
    root = pattern_path or hp.Train.Pattern_Path
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, file_Name).replace("\\", "/"), "wb") as f:
        pickle.dump({"Token": token, "Mel": np.asarray(mel, np.float32), "Text": text, "Dataset": dataset}, f, protocol=2)


def Pattern_File_Generate(path, text, token_Index_Dict, dataset, file_Prefix="", range_Ignore=False, device="cuda"):
    """Pattern_Generate.py:60-82 for one (wav, text) pair; returns the pickle name or None when the utterance is skipped."""
    text = Text_Filtering(text)
    mel = Mel_Generate(path, range_Ignore, device=device) if text is not None else None
    if mel is None:
        return None
    name = "{}.{}{}.PICKLE".format(dataset, file_Prefix, os.path.splitext(os.path.basename(path))[0]).upper()
    Pattern_File_Write(name, text, mel, token_Index_Dict, dataset)
    return name


def Metadata_Generate(token_Index_Dict=None, pattern_path=None):
    """Pattern_Generate.py:245-274."""
    root = pattern_path or hp.Train.Pattern_Path
    md = {"Token_Index_Dict": token_Index_Dict or _Feeder.load_token_dict(), "Spectrogram_Dim": hp.Sound.Spectrogram_Dim, "Mel_Dim": hp.Sound.Mel_Dim,
          "Frame_Shift": hp.Sound.Frame_Shift, "Frame_Length": hp.Sound.Frame_Length, "Sample_Rate": hp.Sound.Sample_Rate,
          "File_List": [], "Token_Length_Dict": {}, "Mel_Length_Dict": {}, "Dataset_Dict": {}}
    meta = hp.Train.Metadata_File.upper()
    for r, _, files in os.walk(root):
        for file in sorted(files):
            if file == meta:
                continue
            try:
                with open(os.path.join(r, file).replace("\\", "/"), "rb") as f:
                    pd = pickle.load(f)
                md["Token_Length_Dict"][file] = pd["Token"].shape[0]
                md["Mel_Length_Dict"][file] = pd["Mel"].shape[0]
                md["Dataset_Dict"][file] = pd["Dataset"]
                md["File_List"].append(file)
            except Exception:
                print("File '{}' is not correct pattern file. This file is ignored.".format(file))
    with open(os.path.join(root, meta).replace("\\", "/"), "wb") as f:
        pickle.dump(md, f, protocol=2)
    print("Metadata generate done.")
    return md
