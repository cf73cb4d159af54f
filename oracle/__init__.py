"""CPU oracle for the multi-speaker Tacotron2 hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is a CPU restatement (NumPy fp64 / torch-CPU) of the
reference algorithm, written from the reference's source semantics with every function citing the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``multi_speaker_tts_amd`` never does, and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference is Python on TensorFlow 1.12-1.13 + librosa<0.8.  Neither is
installed here (no network), the reference has no tests, fixtures or golden vectors, and the
arithmetic lives in those third-party libraries (TensorFlow 1.x: tf.layers.*, tf.contrib.seq2seq
AttentionWrapper/BahdanauAttention, tf.nn.dynamic_rnn, tf.train.AdamOptimizer; librosa:
stft/filters.mel).  The restatement therefore follows the published behaviour of those libraries
at the reference's call sites.  What IS pinned exactly: the integer contracts derivable from
reference data (Token_Index_Dict.json, Inference_Sentence_in_Train.txt tokenisation, speaker-window
arithmetic, STFT framing constants, LR schedule constants) - see tests/test_kats.py.  Two
independent restatements (NumPy fp64 in ``np_ops`` and torch in ``model``) cross-check each other.
The library semantics that have an independent implementation in this image are additionally pinned
to it (tests/test_cpu_thirdparty_pins.py): scipy.signal.lfilter (the reference's own pre-emphasis
call) / get_window / istft, torch.stft / istft / conv1d('same') / batch_norm / nn.LSTM incl. packed
bidirectional sequences / optim.Adam / loss functionals, transformers' Slaney mel filter bank,
pyarrow's snappy codec.  That narrows what "unpinned" covers - the attention, zoneout, decoder
wiring, GE2E and WaveGlow restatements - it does not lift it: the reference itself never ran here.
"""
