"""Top-level drop-in name of the reference driver module: `from MSTTS_SV import Tacotron2` works unchanged
(reference MSTTS_SV.py:20; the implementation lives in multi_speaker_tts_amd/MSTTS_SV.py)."""
from multi_speaker_tts_amd.MSTTS_SV import *          # noqa: F401,F403
from multi_speaker_tts_amd.MSTTS_SV import Tacotron2, TRAIN_KEYS, INFERENCE_KEYS  # noqa: F401
