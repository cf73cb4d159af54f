"""The C-ABI library loads on a GPU-less host and exports exactly the symbols include/mstts.h declares;
the ctypes table, the header and the library agree; the Hyper_Parameters tree matches the reference's."""
import ctypes
import os
import re

import pytest

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mstts.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mstts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from multi_speaker_tts_amd import build, lib
    path = build.build()
    cdll = ctypes.CDLL(path)
    declared = _header_symbols()
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(cdll, name), "library does not export %s" % name
    assert sorted(lib.SIGNATURES) == declared, set(lib.SIGNATURES) ^ set(declared)
    cdll.mstts_abi_version.restype = ctypes.c_int
    from multi_speaker_tts_amd.lib import ABI_VERSION
    assert cdll.mstts_abi_version() == ABI_VERSION == 5
    # host-only helpers are callable without a GPU
    cdll.mstts_skinny_fwd_splits.restype = ctypes.c_int32
    cdll.mstts_skinny_fwd_splits.argtypes = [ctypes.c_int64, ctypes.c_int64]
    assert cdll.mstts_skinny_fwd_splits(4096, 1792) == 8 and cdll.mstts_skinny_fwd_splits(4096, 80) == 0
    cdll.mstts_skinny_bwd_splits.restype = ctypes.c_int32
    cdll.mstts_skinny_bwd_splits.argtypes = [ctypes.c_int64, ctypes.c_int64]
    assert cdll.mstts_skinny_bwd_splits(1792, 4096) == 8


def test_library_reads_no_environment_variable():
    """include/mstts.h: "the library reads NO environment variable" - the development switches are setters (mstts_gemm_*), and the Python
    binding maps its MSTTS_GEMM_* variables onto them in lib.load().  Every source the library is compiled from is searched."""
    import glob
    import re
    csrc = os.path.join(ROOT, "multi_speaker_tts_amd", "csrc")
    hits = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inc"))):
        for n, line in enumerate(open(f), 1):
            if re.search(r"\b(secure_)?getenv\s*\(", line):
                hits.append("%s:%d" % (os.path.basename(f), n))
    assert not hits, hits
    from multi_speaker_tts_amd import lib
    for setter in ("mstts_gemm_split3", "mstts_gemm_split_big", "mstts_gemm_bf16_big", "mstts_gemm_bf16_autocut", "mstts_gemm_big_min_workgroups", "mstts_gemm_tail_split"):
        assert setter in lib.SIGNATURES


def test_struct_layouts_match_header_sizes():
    """ctypes.Structure sizes == sizeof in C (compiled with the same header)."""
    import subprocess, tempfile
    from multi_speaker_tts_amd import lib
    names = {"mstts_gemm_desc": lib.GemmDesc, "mstts_lstm_point_fwd_desc": lib.LstmPointFwd, "mstts_lstm_point_bwd_desc": lib.LstmPointBwd,
             "mstts_lsa_const": lib.LsaConst, "mstts_lstm_seq_fwd_desc": lib.LstmSeqFwd, "mstts_lstm_seq_bwd_desc": lib.LstmSeqBwd,
             "mstts_decoder_train_desc": lib.DecoderTrain, "mstts_decoder_train_bwd_desc": lib.DecoderTrainBwd,
             "mstts_decoder_infer_desc": lib.DecoderInfer}
    src = '#include <stdio.h>\n#include "mstts.h"\nint main(){' + "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    for line in out.strip().splitlines():
        n, sz = line.split()
        assert ctypes.sizeof(names[n]) == int(sz), (n, ctypes.sizeof(names[n]), sz)


def test_hparams_tree():
    from multi_speaker_tts_amd import Hyper_Parameters as hp
    assert hp.Sound.Sample_Rate == 16000 and hp.Sound.Mel_Dim == 80 and hp.Sound.Max_Abs_Mel == 4
    assert hp.Encoder.Conv.Kernel_Size == 5 and hp.Encoder.BiLSTM.Cell_Size == 256 and hp.Attention.Conv.Kernel_Size == 31
    assert hp.Decoder.PreNet.Use_Dropout is True and hp.Decoder.LSTM.Max_Inference_Length == 1000
    assert hp.Train.ADAM.Epsilon == 1e-6 and hp.Train.Learning_Rate.Decay_Step == 10000 and hp.Train.Use_Wav_Length_Range == (500, 9000)
    assert hp.Speaker_Embedding.Inference.Sample_Nums == 5 and hp.Taco1_Mel_to_Spect.ConvBank.Max_Kernel_Size == 8
    assert hp.WaveGlow.Train.Max_Signal_Length == 8000 and hp.Use_Vocoder == "Taco1_Mel_to_Spect"
    assert set(hp.Decoder.values()) == {"PreNet", "LSTM", "Conv"}


def test_variable_table_counts():
    from multi_speaker_tts_amd import params as PP
    from oracle import model as OM
    d = PP.Dims()
    t = PP.variable_table(d)
    n_train = sum(int(np.prod(s)) for n, s, _ in t if PP.is_trainable(n))
    assert n_train == 30278977
    assert [(n, tuple(s)) for n, s, _ in t] == [(n, tuple(s)) for n, s, _ in OM.param_specs(OM.Dims())]
    a, b = PP.initial_values(PP.Dims(emb=8, enc_conv_ch=8), 3), OM.init_params(OM.Dims(emb=8, enc_conv_ch=8), 3)
    assert all(np.allclose(a[k], b[k].astype(np.float32)) for k in a)


def test_top_level_drop_in_names():
    """`from MSTTS_SV import Tacotron2` / `import Hyper_Parameters as hp` work unchanged from the repo root."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    hp_top = importlib.import_module("Hyper_Parameters")
    from multi_speaker_tts_amd import Hyper_Parameters as hp_pkg
    assert hp_top is hp_pkg and hp_top.Sound.Mel_Dim == 80
    mod = importlib.import_module("MSTTS_SV")
    from multi_speaker_tts_amd.MSTTS_SV import Tacotron2
    assert mod.Tacotron2 is Tacotron2
    import inspect
    sig = inspect.signature(Tacotron2.__init__)
    assert list(sig.parameters)[:2] == ["self", "is_Training"] and sig.parameters["is_Training"].default is False
    for name in ("Restore", "Train", "Inference"):
        assert callable(getattr(Tacotron2, name))
    assert list(inspect.signature(Tacotron2.Inference).parameters)[:4] == ["self", "path_List", "text_List", "file_Prefix"]


def test_split_k_choices_of_the_train_step():
    """Host logic: the reduction split of the weight-gradient products (engine._split_k).  The cases are the shapes of one config-2
    step with the split that was measured best on the GPU (tools/gemm_step_profile.py): long reductions want three workgroups per CU,
    shapes that already fill the chip whole rounds."""
    from multi_speaker_tts_amd.engine import _split_k
    assert _split_k(2560, 512, 25632) == 9          # postnet convolution gradients: 80 tiles -> 720 = 3 per CU (sk 3: 91, sk 9: 117 TFLOP/s)
    assert _split_k(1792, 4096, 25632) == 4         # dw0f: 448 tiles -> 7 per CU exactly
    assert _split_k(2048, 4096, 25632) in (1, 2)    # dW1: 512 tiles = 2 per CU either way
    assert _split_k(2560, 512, 4096) == 3           # encoder convolution gradients (short reduction: the round model)
    for M, N, K in ((256, 4096, 25632), (1024, 128, 25632), (80, 256, 25632), (512, 1024, 4096)):
        sk = _split_k(M, N, K)
        assert 1 <= sk <= 64 and (K < 16384 or K // sk >= 400)          # (a handful of tiles may be split up to 64 ways: one workgroup per CU)


def test_deterministic_gemm_context_nests(monkeypatch):
    """Host logic of lib.deterministic_gemm: the per-thread switch goes on at the outermost entry and off at the outermost exit, also through
    the decorator form and across an exception; another thread starts from its own depth 0."""
    import threading
    from multi_speaker_tts_amd import lib
    L = lib.load()
    calls = []
    real = L.mstts_gemm_deterministic
    monkeypatch.setattr(L, "mstts_gemm_deterministic", lambda on: calls.append((threading.get_ident(), on)) or real(on))
    me = threading.get_ident()
    with lib.deterministic_gemm():
        with lib.deterministic_gemm():
            pass

        @lib.deterministic_gemm()
        def f():
            return 7
        assert f() == 7
        assert calls == [(me, 1)]
    assert calls == [(me, 1), (me, 0)]
    with pytest.raises(ValueError):
        with lib.deterministic_gemm():
            raise ValueError("x")
    assert calls[-2:] == [(me, 1), (me, 0)]
    seen = []

    def other():
        with lib.deterministic_gemm():
            seen.append([c for c in calls if c[0] == threading.get_ident()])
    with lib.deterministic_gemm():
        t = threading.Thread(target=other); t.start(); t.join()
    assert seen and seen[0] == [(seen[0][0][0], 1)]
