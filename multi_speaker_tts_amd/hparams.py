"""TF-free stand-in for ``tf.contrib.training.HParams`` plus the attribute tree of the
reference's ``Hyper_Parameters.py`` (reference: Hyper_Parameters.py:4-241).

The reference keeps its configuration as nested ``HParams`` objects that are read with dot
access at graph-build time.  TensorFlow 1.x does not exist on the MI355X box, so the same tree
(same attribute names, same values) is rebuilt here from one plain nested mapping.
"""
from __future__ import annotations


class HParams:
    """Dot-access namespace with the small part of the HParams API the reference uses."""

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)

    def values(self):
        return {k: (v.values() if isinstance(v, HParams) else v) for k, v in self.__dict__.items()}

    def set_hparam(self, name, value):
        if not hasattr(self, name):
            raise KeyError(name)
        setattr(self, name, value)

    def __contains__(self, name):
        return name in self.__dict__

    def __repr__(self):
        return "HParams(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


def _tree(node):
    if isinstance(node, dict):
        return HParams(**{k: _tree(v) for k, v in node.items()})
    return node


_LR = lambda init, mn, step, rate, start=None: dict(
    {"Initial": init, "Min": mn}, **({} if start is None else {"Decay_Start_Step": start}),
    **{"Decay_Step": step, "Decay_Rate": rate})
_ADAM = lambda eps: {"Beta1": 0.9, "Beta2": 0.999, "Epsilon": eps}
_CONV = lambda n, k, ch: {"Nums": n, "Kernel_Size": k, "Stride": 1, "Channel": ch, "Dropout_Rate": 0.5}

# One mapping, section per reference block (Hyper_Parameters.py line ranges in comments).
TREE = {
    "Sound": {  # :4-11
        "Sample_Rate": 16000, "Spectrogram_Dim": 1025, "Mel_Dim": 80, "Max_Abs_Mel": 4,
        "Frame_Shift": 12.5, "Frame_Length": 50},
    "Encoder": {  # :13-30
        "Embedding": {"Token_Size": 42, "Embedding_Size": 512},
        "Conv": _CONV(3, 5, 512),
        "BiLSTM": {"Nums": 1, "Cell_Size": 256, "Zoneout_Rate": 0.1}},
    "Attention": {  # :32-40
        "Memory_Size": 128,
        "Conv": {"Kernel_Size": 31, "Stride": 1, "Channel": 32, "Dropout_Rate": 0.5}},
    "Decoder": {  # :42-62
        "PreNet": {"Nums": 2, "Size": 256, "Use_Dropout": True, "Dropout_Rate": 0.5},
        "LSTM": {"Nums": 2, "Cell_Size": 1024, "Zoneout_Rate": 0.1, "Max_Inference_Length": 1000},
        "Conv": _CONV(5, 5, 512)},
    "Train": {  # :64-91
        "Pre_Step": 0, "Use_Pre_in_Main_Train": False,
        "Pattern_Path": "E:/MSTTS_SV.Data", "Metadata_File": "METADATA.PICKLE",
        "Batch_Size": 32, "Pattern_Sorting_by_Mel_Length": True,
        "Use_Wav_Length_Range": (500, 9000),
        "Pre_Train_Dataset_List": ["LJ"], "Main_Train_Dataset_List": ["VCTK", "TIMIT"],
        "Max_Pattern_Queue": 20,
        "Learning_Rate": _LR(1e-3, 1e-5, 10000, 0.5, start=0),
        "Weight_Regularization_Rate": 1e-6, "ADAM": _ADAM(1e-6), "Use_L1_Loss": True,
        "Inference_Timing": 1000, "Checkpoint_Save_Timing": 1000},
    "Speaker_Embedding": {  # :93-131
        "Embedding_Size": 256,
        "LSTM": {"Nums": 3, "Cell_Size": 256, "Zoneout_Rate": 0.1, "Use_Residual": True},
        "Inference": {"Sample_Nums": 5, "Mel_Frame": 64, "Overlap_Frame": 32,
                      "Max_Embedding_per_Batch": 128},
        "Checkpoint_Path": "E:/Speaker_Embedding/Checkpoint",
        "Train": {
            "Pattern_Path": "E:/Speaker_Embedding.Data", "Metadata_File": "METADATA.PICKLE",
            "Batch_Speaker": 32, "Batch_per_Speaker": 10, "Max_Pattern_Queue": 20,
            "Frame_Range": (140, 180), "Loss_Calc_Method": "Softmax",
            "Learning_Rate": _LR(1e-3, 1e-5, 10000, 0.5), "ADAM": _ADAM(1e-8),
            "Inference_Path": "E:/MSTTS_Checkpoints/Speaker_Embedding_Checkpoint",
            "Inference_Timing": 1000, "Checkpoint_Save_Timing": 1000}},
    "Taco1_Mel_to_Spect": {  # :133-191
        "ConvBank": {
            "Nums": 1, "Max_Kernel_Size": 8, "Stride": 1, "Channel": 128,
            "Pooling": {"Size": 2, "Stride": 1},
            "Projection1": {"Kernel_Size": 3, "Stride": 1, "Channel": 256},
            "Projection2": {"Kernel_Size": 3, "Stride": 1, "Channel": 80},
            "Dropout_Rate": 0.5},
        "Highway": {"Nums": 4},
        "BiRNN": {"Nums": 1, "Cell_Size": 128, "Zoneout_Rate": 0.1},
        "Griffin_Lim_Iteration": 100,
        "Checkpoint_Path": "E:/MSTTS_Checkpoints/Mel_to_Spect_Checkpoint",
        "Train": {
            "Pattern_Path": "E:/Taco1_Mel_to_Spect.Data/", "Metadata_File": "METADATA.PICKLE",
            "Batch_Size": 128, "Pattern_Sorting_by_Length": True, "Max_Mel_Length": 1000,
            "Max_Pattern_Queue": 20,
            "Learning_Rate": _LR(1e-3, 1e-5, 100, 0.5, start=50000),
            "Weight_Regularization_Rate": 1e-6, "ADAM": _ADAM(1e-6),
            "Inference_Timing": 1000, "Checkpoint_Save_Timing": 1000,
            "Inference": {"Path": "E:/MtS(20190817)", "Batch_Size": 128}}},
    "WaveGlow": {  # :195-236
        "Flows": 12, "Groups": 8, "Early_Every": 4, "Early_Size": 2,
        "Upsample": {"Kernel_Size": 1024, "Strides": 256},
        "WaveNet": {"Layers": 8, "Channels": 512, "Kernel_Size": 3},
        "Export_Sample_Rate": 22050,
        "Checkpoint_Path": "E:/MSTTS_SV_for_WaveGlow_Server/Checkpoint",
        "Train": {
            "Pattern_Path": "E:/Multi_Speaker_TTS.Raw_Data/VCTK/wav48",
            "Max_Signal_Length": 16000 // 2, "Batch_Size": 4, "Max_Pattern_Queue": 20,
            "Learning_Rate": _LR(1e-3, 1e-5, 100000, 0.5), "ADAM": _ADAM(1e-8),
            "Inference_Timing": 1000, "Checkpoint_Save_Timing": 1000},
        "Inference": {"Path": "E:/WaveGlow", "Mel_Split_Length": 40, "Batch_Size": 4}},
}

# Module-level scalars of the reference (Hyper_Parameters.py:238-241).
SCALARS = {
    "Use_Vocoder": "Taco1_Mel_to_Spect",
    "Inference_Path": "E:/MSTTS_Test(20190827)",
    "Checkpoint_Path": "E:/MSTTS_Checkpoints/Multi_Speaker_TTS_Checkpoint",
}


def build():
    """Fresh copy of the whole tree: {section name: HParams} plus the scalars."""
    out = {name: _tree(node) for name, node in TREE.items()}
    out.update(SCALARS)
    return out
