"""Drop-in for the reference's ``Hyper_Parameters`` module (``import Hyper_Parameters as hp``).

Same attribute names and values (reference Hyper_Parameters.py:4-241); the tree itself lives in
``hparams.py`` because ``tf.contrib.training.HParams`` is not available without TensorFlow 1.x.
"""
from .hparams import build as _build

globals().update(_build())
