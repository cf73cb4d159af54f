"""Top-level drop-in name of the reference's hyper-parameter module (`import Hyper_Parameters as hp`, reference
Hyper_Parameters.py:4-241).  It IS the package's module object, so an edit made through either name is seen by the engines."""
import sys

from multi_speaker_tts_amd import Hyper_Parameters as _hp

sys.modules[__name__] = _hp
