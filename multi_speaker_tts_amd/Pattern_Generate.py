"""The on-disk training-pattern format of the reference (Pattern_Generate.py:14-76,245-274): one pickle (protocol 2) per
utterance holding {'Token': int32[T_tok], 'Mel': float32[T_mel, 80], 'Text': str, 'Dataset': str}, named
'<DATASET>.<prefix><wav basename>.PICKLE', plus METADATA.PICKLE with the hyper parameters the set was made with and the
per-file lengths, the corpus walkers of Pattern_Generate.py:115-243 (LJSpeech, VCTK, LibriSpeech, TEDLIUM, TIMIT directory
layouts -> (audio path, filtered text) pairs) and the reference's command line (`-lj -vctk -ls -tl -timit -all`, :277-285).
Mel extraction runs on the GPU (mstts_stft_mel); file decoding is scipy / the SPHERE reader in Feeder.read_sphere.
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


def Mel_Generate(path, spectral_Subtract=False, range_Ignore=False, device="cuda"):
    """Pattern_Generate.py:33-58 (same positional order): load at hp.Sound.Sample_Rate, trim (top_db 15, librosa's default
    2048/512 frames), scale by 0.99, reject by duration, mel through the HIP STFT kernel."""
    from . import Audio
    sig = _Feeder.load_wav(path, frame=2048, hop=512)
    ms = sig.shape[0] / hp.Sound.Sample_Rate * 1000
    if not range_Ignore and (ms < hp.Train.Use_Wav_Length_Range[0] or ms > hp.Train.Use_Wav_Length_Range[1]):
        return None
    return np.transpose(Audio.melspectrogram(y=sig, num_freq=hp.Sound.Spectrogram_Dim, frame_shift_ms=hp.Sound.Frame_Shift,
                                             frame_length_ms=hp.Sound.Frame_Length, num_mels=hp.Sound.Mel_Dim, sample_rate=hp.Sound.Sample_Rate,
                                             max_abs_value=hp.Sound.Max_Abs_Mel, spectral_subtract=spectral_Subtract,
                                             device=device)).astype(np.float32)


def Pattern_File_Write(file_Name, text, mel, token_Index_Dict, dataset, pattern_path=None):
    """Pattern_Generate.py:66-76."""
    token = np.array([token_Index_Dict[letter] for letter in text]).astype(np.int32)
    root = pattern_path or hp.Train.Pattern_Path
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, file_Name).replace("\\", "/"), "wb") as f:
        pickle.dump({"Token": token, "Mel": np.asarray(mel, np.float32), "Text": text, "Dataset": dataset}, f, protocol=2)


def Pattern_File_Generate(path, text, token_Index_Dict, dataset, spectral_Subtract=False, file_Prefix="", display_Prefix="", range_Ignore=False,
                          device="cuda"):
    """Pattern_Generate.py:60-82 for one (wav, text) pair; returns the pickle name or None when the utterance is skipped."""
    text = Text_Filtering(text)
    mel = Mel_Generate(path, spectral_Subtract, range_Ignore, device=device) if text is not None else None
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


# ---------------------------------------------------------------------------------------------------------------------
# corpus walkers (Pattern_Generate.py:115-243): each returns (path list, {path: text}) - for TEDLIUM {path: [(start, end, text)]}
# ---------------------------------------------------------------------------------------------------------------------
using_Extension = [x.upper() for x in [".wav", ".m4a", ".flac"]]


def _slash(*parts):
    return os.path.join(*parts).replace("\\", "/")


def _audio_files(root_dir):
    """Every file under root_dir (sorted walk) whose extension is one the reference accepts."""
    for root, dirs, files in os.walk(root_dir):
        dirs.sort()
        for name in sorted(files):
            if os.path.splitext(name)[1].upper() in using_Extension:
                yield root.replace("\\", "/"), name


def LJ_Info_Load(lj_Path):
    """LJSpeech: metadata.csv rows `id|raw text|normalised text`; wavs/<id>.wav (Pattern_Generate.py:115-140)."""
    paths, texts = [], {}
    with open(_slash(lj_Path, "metadata.csv"), "r", encoding="utf-8-sig") as f:
        rows = [[x.strip() for x in line.split("|")] for line in f.readlines() if line.strip()]
    for row in rows:
        wav = _slash(lj_Path, "wavs", "{}.wav".format(row[0]))
        text = Text_Filtering(row[2]) if len(row) > 2 else None
        if text is None or not os.path.exists(wav):
            continue
        if wav not in texts:
            paths.append(wav)
        texts[wav] = text
    print("LJ info generated.")
    return paths, texts


def VCTK_Info_Load(vctk_Path):
    """VCTK: wav48/<speaker>/<utt>.wav with its transcript at txt/<speaker>/<utt>.txt (Pattern_Generate.py:142-164)."""
    wav_root, txt_root = _slash(vctk_Path, "wav48"), _slash(vctk_Path, "txt")
    paths, texts = [], {}
    for root, name in _audio_files(wav_root):
        wav = _slash(root, name)
        txt = os.path.splitext(wav.replace(wav_root, txt_root, 1))[0] + ".txt"
        if not os.path.exists(txt):
            continue
        with open(txt, "r") as f:
            text = Text_Filtering(f.read().strip())
        if text is None:
            continue
        paths.append(wav)
        texts[wav] = text
    print("VCTK info generated.")
    return paths, texts


def LS_Info_Load(ls_Path):
    """LibriSpeech: <speaker>/<chapter>/<speaker>-<chapter>.trans.txt lists `<utt id> <TEXT>`; audio beside it (Pattern_Generate.py:166-196)."""
    paths, texts = [], {}
    for root, name in _audio_files(ls_Path):
        speaker, chapter = root.split("/")[-2:]
        trans = _slash(root, "{}-{}.trans.txt".format(speaker, chapter))
        if not os.path.exists(trans):
            continue
        table = {}
        with open(trans, "r") as f:
            for line in f.readlines():
                parts = line.strip().split(" ")
                table[parts[0]] = " ".join(parts[1:])
        key = os.path.splitext(name)[0]
        text = Text_Filtering(table[key]) if key in table else None
        if text is None:
            continue
        wav = _slash(root, name)
        paths.append(wav)
        texts[wav] = text
    print("LS info generated.")
    return paths, texts


def TL_Info_Load(tl_Path):
    """TEDLIUM: sph/<talk>.sph with segments in stm/<talk>.stm (`talk ch speaker start end <tags> words...`); segments holding <UNK>
    are dropped (Pattern_Generate.py:198-222)."""
    sph_root, stm_root = _slash(tl_Path, "sph"), _slash(tl_Path, "stm")
    paths, segs = [], {}
    for root, dirs, files in os.walk(sph_root):
        dirs.sort()
        for name in sorted(files):
            sph = _slash(root, name)
            paths.append(sph)
            stm = os.path.splitext(sph.replace(sph_root, stm_root, 1))[0] + ".stm"
            if not os.path.exists(stm):
                continue
            segs[sph] = []
            with open(stm, "r", encoding="utf-8-sig") as f:
                for words in [x.strip().upper().split(" ") for x in f.readlines() if x.strip()]:
                    if "<UNK>" in words:
                        continue
                    text = Text_Filtering(" ".join(words[6:]).replace(" '", "'"))
                    if text is not None:
                        segs[sph].append((float(words[3]), float(words[4]), text))
    print("TL info generated.")
    return paths, segs


def TIMIT_Info_Load(timit_Path):
    """TIMIT: <dialect>/<speaker>/<utt>.WAV (SPHERE) with <utt>.TXT = `start end words...` (Pattern_Generate.py:224-243)."""
    paths, texts = [], {}
    for root, name in _audio_files(timit_Path):
        wav = _slash(root, name)
        base, ext = os.path.splitext(wav)
        txt = base + (".TXT" if ext.isupper() else ".txt")
        if not os.path.exists(txt):
            continue
        with open(txt, "r") as f:
            text = Text_Filtering(" ".join(f.read().strip().split(" ")[2:]).strip())
        if text is None:
            continue
        paths.append(wav)
        texts[wav] = text
    print("TIMIT info generated.")
    return paths, texts


def Pattern_File_Generate_from_SPH(path, text_List, token_Index_Dict, dataset, spectral_Subtract=False, display_Prefix="", range_Ignore=False,
                                   device="cuda"):
    """Pattern_Generate.py:80-113: one pattern per (start, end, text) segment of a SPHERE recording, named
    <DATASET>.<basename>.<index>.PICKLE.  Returns the names written.  Two things the reference does here are kept on purpose, so that
    the TEDLIUM pattern set comes out identical: it calls Mel_Generate with its DEFAULTS (`:91` - the spectral_Subtract / range_Ignore
    arguments of this function are accepted and ignored, also under `-all`), and it stops at the FIRST segment the length filter
    rejects (`:92-94`, a `return`, not a `continue`): later segments of that recording produce no pattern."""
    from scipy.io import wavfile
    import tempfile
    names = []
    for index, (start, end, text) in enumerate(text_List):
        rate, data = _Feeder.read_sphere(path, start, end)
        with tempfile.NamedTemporaryFile(suffix=".wav", delete=False) as tf:
            tmp = tf.name
        try:
            wavfile.write(tmp, rate, data)
            mel = Mel_Generate(tmp, device=device)
        finally:
            os.remove(tmp)
        if mel is None:
            return names
        name = "{}.{}.{}.PICKLE".format(dataset, os.path.splitext(os.path.basename(path))[0], index).upper()
        Pattern_File_Write(name, text, mel, token_Index_Dict, dataset)
        names.append(name)
    return names


def main(argv=None, device="cuda"):
    """The reference's command line (Pattern_Generate.py:277-404): walk the given corpora, write one pattern per utterance into
    hp.Train.Pattern_Path, then METADATA.PICKLE.  `-all`: keep utterances outside hp.Train.Use_Wav_Length_Range as well."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("-lj", "--lj_path", required=False)
    ap.add_argument("-vctk", "--vctk_path", required=False)
    ap.add_argument("-ls", "--ls_path", required=False)
    ap.add_argument("-tl", "--tl_path", required=False)
    ap.add_argument("-timit", "--timit_path", required=False)
    ap.add_argument("-all", "--all_save", action="store_true")
    args = ap.parse_args(argv)
    token_Index_Dict = _Feeder.load_token_dict()
    jobs = []                                            # (dataset, path, text or segment list)
    # per corpus, as the reference submits them (Pattern_Generate.py:318-404): LibriSpeech (and TEDLIUM, where the flag is then dropped)
    # with spectral subtraction, the others without; TIMIT names carry the speaker directory, because its utterance names (SA1, SA2,
    # SX...) repeat from speaker to speaker
    spectral = {"LJ": False, "VCTK": False, "LS": True, "TL": True, "TIMIT": False}
    for dataset, root, loader in (("LJ", args.lj_path, LJ_Info_Load), ("VCTK", args.vctk_path, VCTK_Info_Load), ("LS", args.ls_path, LS_Info_Load),
                                  ("TL", args.tl_path, TL_Info_Load), ("TIMIT", args.timit_path, TIMIT_Info_Load)):
        if root is None:
            continue
        paths, table = loader(root)
        jobs += [(dataset, p, table[p]) for p in paths if p in table]
    if not jobs:
        raise ValueError("Total pattern count is zero.")
    os.makedirs(hp.Train.Pattern_Path, exist_ok=True)
    written = 0
    for i, (dataset, path, what) in enumerate(jobs):
        if dataset == "TL":
            names = Pattern_File_Generate_from_SPH(path, what, token_Index_Dict, dataset, spectral[dataset], range_Ignore=args.all_save, device=device)
        else:
            prefix = "{}.".format(path.split("/")[-2]) if dataset == "TIMIT" else ""
            name = Pattern_File_Generate(path, what, token_Index_Dict, dataset, spectral[dataset], prefix, range_Ignore=args.all_save, device=device)
            names = [name] if name else []
        written += len(names)
        print("[{} {:05d}/{:05d}]".format(dataset, i, len(jobs)), path, "->", ", ".join(names) if names else "Ignored because of length.")
    Metadata_Generate(token_Index_Dict)
    return written


if __name__ == "__main__":
    main()
