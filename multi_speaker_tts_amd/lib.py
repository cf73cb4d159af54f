"""ctypes binding of libmstts_hip.so (the C ABI in include/mstts.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, importing any
compute entry point raises.  PyTorch only provides device memory (``tensor.data_ptr()``) and the
current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmstts_hip.so")

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3

vp, i64, i32, f32, u64, u32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint32


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp),
                ("M", i64), ("N", i64), ("K", i64), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("trans_a", i32), ("trans_b", i32), ("win_T", i32), ("win_C", i32), ("win_pad", i32),
                ("act", i32), ("accumulate", i32), ("split_k", i32),
                ("batch", i64), ("stride_a", i64), ("stride_b", i64), ("stride_c", i64), ("alpha", f32), ("win_dil", i32)]


class LstmPointFwd(C.Structure):
    _fields_ = [("B", i64), ("H", i64), ("gates_h", vp), ("gates_parts", i32), ("gates_pstride", i64), ("xw", vp), ("xw_sb", i64), ("xw_st", i64), ("bias", vp),
                ("c_prev", vp), ("h_prev", vp), ("h_prev_ld", i64), ("zc", vp), ("zh", vp), ("zoneout", f32),
                ("lengths", vp), ("step", i32), ("reverse", i32), ("residual", vp), ("res_sb", i64), ("res_st", i64),
                ("out", vp), ("out_sb", i64), ("out_st", i64), ("c_next", vp), ("h_next", vp), ("h_next_ld", i64),
                ("acts_out", vp), ("c_raw", vp)]


class LstmPointBwd(C.Structure):
    _fields_ = [("B", i64), ("H", i64), ("d_out", vp), ("dout_sb", i64), ("dout_st", i64), ("dout_parts", i32), ("dout_pstride", i64),
                ("d_out2", vp), ("dout2_parts", i32), ("dout2_pstride", i64),
                ("d_c_state", vp), ("d_h_state", vp), ("d_h_state2", vp), ("dhs2_ld", i64), ("dhs2_parts", i32), ("dhs2_pstride", i64),
                ("acts", vp), ("c_raw", vp), ("c_prev", vp), ("zc", vp), ("zh", vp), ("zoneout", f32),
                ("lengths", vp), ("step", i32), ("reverse", i32), ("dgates", vp), ("dgates_pos", vp),
                ("dgp_sb", i64), ("dgp_st", i64), ("d_c_prev", vp), ("d_h_prev", vp), ("dq", vp), ("wq_t", vp), ("A", i64), ("dq_bf16", i32)]


class CellPackedDst(C.Structure):
    _fields_ = [("base", vp), ("K", i64), ("col0", i64), ("bf16", i32)]


class LsaPrenet(C.Structure):
    _fields_ = [("w0", vp), ("b0", vp), ("w1", vp), ("b1", vp), ("m0", vp), ("m1", vp), ("inv_keep", f32), ("P", i32),
                ("out", vp), ("out_ld", i64), ("out_p", CellPackedDst)]


class CellFwd(C.Structure):
    _fields_ = [("B", i64), ("H", i64), ("K", i64), ("Xp", vp), ("Wp", vp), ("xw", vp), ("xw_ld", i64), ("bias", vp),
                ("c_prev", vp), ("h_prev", vp), ("h_prev_ld", i64), ("zc", vp), ("zh", vp), ("zoneout", f32),
                ("out", vp), ("out_ld", i64), ("c_next", vp), ("h_next", vp), ("h_next_ld", i64), ("acts", vp), ("c_raw", vp),
                ("out_p", CellPackedDst), ("h_next_p", CellPackedDst),
                ("lengths", vp), ("step", i32), ("reverse", i32), ("xw_st", i64), ("out_st", i64), ("bf16", i32)]


class LsaConst(C.Structure):
    _fields_ = [("B", i64), ("T", i64), ("A", i64), ("M", i64), ("KS", i64), ("CH", i64),
                ("keys", vp), ("values", vp), ("lengths", vp),
                ("conv_k", vp), ("conv_b", vp), ("dense_k", vp), ("score_w", vp), ("score_b", vp), ("loc_k", vp), ("loc_b", vp), ("loc_kt", vp)]


class LstmSeqFwd(C.Structure):
    _fields_ = [("B", i64), ("T", i64), ("H", i64), ("xw", vp), ("wh", vp), ("wh_ld", i64), ("lengths", vp),
                ("reverse", i32), ("zoneout", f32), ("zc", vp), ("zh", vp), ("residual", vp),
                ("out", vp), ("out_sb", i64), ("out_st", i64),
                ("c_hist", vp), ("h_hist", vp), ("acts", vp), ("c_raw", vp), ("gates_ws", vp), ("wh_p", vp), ("h_p", vp)]


class LstmSeqBwd(C.Structure):
    _fields_ = [("B", i64), ("T", i64), ("H", i64), ("wh", vp), ("wh_ld", i64), ("lengths", vp), ("reverse", i32),
                ("zoneout", f32), ("zc", vp), ("zh", vp), ("d_out", vp), ("dout_sb", i64), ("dout_st", i64),
                ("c_hist", vp), ("acts", vp), ("c_raw", vp), ("dgates_step", vp), ("dgates_pos", vp), ("ws", vp)]


class DecoderTrain(C.Structure):
    _fields_ = [("B", i64), ("S", i64), ("H", i64), ("P", i64), ("lsa", LsaConst),
                ("xw0", vp), ("w0f", vp), ("w1", vp), ("b1", vp), ("wq", vp),
                ("zc0", vp), ("zh0", vp), ("zc1", vp), ("zh1", vp), ("zoneout", f32),
                ("in0", vp), ("in1", vp), ("pj", vp), ("c0", vp), ("c1", vp),
                ("acts0", vp), ("acts1", vp), ("craw0", vp), ("craw1", vp),
                ("q_hist", vp), ("align_hist", vp), ("cum_hist", vp), ("gates_ws", vp), ("energy_ws", vp), ("q_ws", vp), ("chains", i32),
                ("bf_w0f_f", vp), ("bf_w1_f", vp), ("bf_wq_f", vp), ("bf_w0f_b", vp), ("bf_w1_b", vp), ("bf_wq_b", vp),
                ("w0p", vp), ("w1p", vp), ("w0p16", vp), ("w1p16", vp), ("w0f_bp", vp), ("w1_bp", vp), ("wq_bp", vp), ("wq_t", vp), ("act_p", vp),
                ("energy_ws_floats", i64)]


class PersistDesc(C.Structure):
    _fields_ = [("w0pk", vp), ("w1pk", vp), ("wqpk", vp), ("xch", vp), ("ctrl", vp), ("stamps", vp), ("opk", vp), ("selftest_fail_step", i32), ("near_xcd", i32), ("pre", vp), ("b0", vp), ("recurrent_bf16", i32)]


class PersistInferDesc(C.Structure):
    _fields_ = [("w0pk", vp), ("w1pk", vp), ("wqppk", vp), ("pre0", vp), ("bf", vp), ("u", vp), ("vp", vp), ("bp_pad", vp),
                ("xch", vp), ("ctrl", vp), ("stamps", vp), ("selftest_fail_step", i32), ("near_xcd", i32)]


class DecoderTrainBwd(C.Structure):
    _fields_ = [("fwd", C.POINTER(DecoderTrain)), ("d_pj", vp), ("dg0", vp), ("dg1", vp), ("dq_hist", vp),
                ("de_hist", vp), ("d_in0", vp), ("ws", vp)]


class DecoderInfer(C.Structure):
    _fields_ = [("B", i64), ("H", i64), ("P", i64), ("n_mel", i64), ("Smax", i64), ("lsa", LsaConst),
                ("pw0", vp), ("pb0", vp), ("pw1", vp), ("pb1", vp), ("pm0", vp), ("pm1", vp), ("prenet_keep", f32),
                ("wx0", vp), ("b0", vp), ("w0f", vp), ("w1", vp), ("b1", vp), ("wq", vp), ("wproj", vp), ("bproj", vp),
                ("zoneout", f32), ("in0", vp), ("in1", vp), ("pj", vp), ("c0", vp), ("c1", vp), ("cum", vp),
                ("pre_ws", vp), ("linear", vp), ("stop", vp), ("align_hist", vp), ("w0s", vp), ("wp_pad", vp), ("w0sp", vp), ("w1p", vp), ("act_p", vp), ("wp_own", vp), ("vp", vp)]


P = C.POINTER
# name -> (restype, argtypes); every symbol include/mstts.h declares
SIGNATURES = {
    "mstts_last_error": (C.c_char_p, []),
    "mstts_abi_version": (i32, []),
    "mstts_debug_park_cus": (i32, [i32, i64, vp, vp]),
    "mstts_gemm_f32": (i32, [P(GemmDesc), vp]),
    "mstts_gemm_tail_split": (i32, [i32]),
    "mstts_gemm_split3": (i32, [i32]),
    "mstts_gemm_deterministic": (i32, [i32]),
    "mstts_gemm_split_big": (i32, [i32]),
    "mstts_gemm_big_min_workgroups": (i32, [i32, i32]),
    "mstts_gemm_bf16_autocut": (i32, [i32]),
    "mstts_persist_status": (i32, [vp, i32, vp, i32, vp, vp]),
    "mstts_gemm_bf16": (i32, [P(GemmDesc), vp]),
    "mstts_gemm_bf16_big": (i32, [i32]),
    "mstts_philox_keep_mask": (i32, [vp, i64, u64, u32, f32, vp]),
    "mstts_philox_keep_mask_rows": (i32, [vp, i64, i64, i64, u64, u32, u64, f32, vp]),
    "mstts_embedding_fwd": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "mstts_embedding_bwd": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "mstts_bn_train_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i64, i64, vp, vp]),
    "mstts_bn_infer_fwd": (i32, [vp, vp, vp, vp, vp, vp, f32, i64, i64, vp]),
    "mstts_bn_train_bwd": (i32, [vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, i64, i64, vp, vp]),
    "mstts_dropout": (i32, [vp, vp, f32, vp, i64, vp]),
    "mstts_relu_dropout_bwd": (i32, [vp, vp, vp, f32, vp, i64, vp]),
    "mstts_colsum": (i32, [vp, i64, i64, i64, vp, i32, vp]),
    "mstts_add": (i32, [vp, vp, vp, i64, vp]),
    "mstts_fill": (i32, [vp, f32, i64, vp]),
    "mstts_copy2d": (i32, [vp, i64, vp, i64, i64, i64, i32, vp]),
    "mstts_maxpool2_same": (i32, [vp, vp, i64, i64, i64, vp]),
    "mstts_highway_combine": (i32, [vp, vp, vp, vp, i64, vp]),
    "mstts_maxpool2_same_bwd": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "mstts_highway_combine_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "mstts_l1_loss_fwd_bwd": (i32, [vp, vp, i64, vp, vp, vp]),
    "mstts_lstm_point_fwd": (i32, [P(LstmPointFwd), vp]),
    "mstts_lstm_point_bwd": (i32, [P(LstmPointBwd), vp]),
    "mstts_cell_fwd_supported": (i32, [i64, i64]),
    "mstts_pack_cell_fwd": (i32, [vp, i64, vp, i64, i64, vp]),
    "mstts_cell_fwd": (i32, [P(CellFwd), vp]),
    "mstts_cell_fwd_bf16_supported": (i32, [i64, i64]),
    "mstts_pack_cell_fwd_bf16": (i32, [vp, i64, vp, i64, i64, vp]),
    "mstts_pack_cell_act_bf16": (i32, [vp, i64, vp, i64, i64, vp]),
    "mstts_cell_fwd_pair": (i32, [P(CellFwd), P(CellFwd), vp]),
    "mstts_cell_act_floats": (i64, [i64, i64]),
    "mstts_pack_cell_act": (i32, [vp, i64, vp, i64, i64, vp]),
    "mstts_lsa_energy_fwd": (i32, [P(LsaConst), vp, i32, i64, vp, vp, vp, vp]),
    "mstts_lsa_context_fwd": (i32, [P(LsaConst), vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "mstts_lsa_step_ws_bytes": (i64, [i64, i64]),
    "mstts_lsa_step_bwd": (i32, [P(LsaConst), vp, i64, vp, i64, i32, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "mstts_lsa_step_fwd": (i32, [P(LsaConst), vp, i32, i64, vp, vp, vp, vp, vp, i64, vp, i64, P(CellPackedDst), vp, C.c_uint32, vp]),
    "mstts_lsa_step_q_supported": (i32, [i64, i64, i64]),
    "mstts_lsa_step_q_ws_bytes": (i64, [i64, i64]),
    "mstts_lsa_step_qp_supported": (i32, [i64, i64, i64, i64]),
    "mstts_lsa_proj_pack_floats": (i64, []),
    "mstts_lsa_proj_pack": (i32, [vp, i64, i64, i64, vp, vp]),
    "mstts_lsa_step_qp_ws_bytes": (i64, [i64, i64]),
    "mstts_lsa_step_prenet_supported": (i32, [i64, i64]),
    "mstts_lsa_step_fwd_qp": (i32, [P(LsaConst), vp, i64, vp, i64, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, P(CellPackedDst),
                                    P(LsaPrenet), vp, C.c_uint32, i32, vp]),
    "mstts_lsa_step_fwd_q": (i32, [P(LsaConst), vp, i64, vp, i64, i32, vp, vp, vp, vp, vp, i64, vp, i64, P(CellPackedDst), vp, C.c_uint32, i32, vp]),
    "mstts_lsa_step_fwd_selftest": (i32, [P(LsaConst), vp, i32, i64, vp, vp, vp, vp, vp, i64, vp, C.c_uint32, i32, vp]),
    "mstts_lsa_dalign_bwd": (i32, [P(LsaConst), vp, i64, vp, i64, i32, i64, vp, vp, vp, vp, vp]),
    "mstts_lsa_denergy_bwd": (i32, [P(LsaConst), vp, vp, vp, vp, vp, vp, vp, vp]),
    "mstts_lsa_param_bwd": (i32, [P(LsaConst), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mstts_lsa_param_bwd_ws_floats": (i64, [i64, i64, i64]),
    "mstts_lsa_fold_location": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "mstts_lsa_filter_by_unit": (i32, [vp, vp, i64, i64, vp]),
    "mstts_lsa_unfold_location_grad": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "mstts_tts_loss_fwd_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i32, f32, vp, vp, vp, vp, vp]),
    "mstts_l2_loss_acc": (i32, [vp, vp, i64, vp, vp]),
    "mstts_shift_frames": (i32, [vp, vp, i64, i64, i64, vp]),
    "mstts_unpack_proj": (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    "mstts_pack_dproj": (i32, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "mstts_speaker_tile": (i32, [vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "mstts_conv_kernel_flip": (i32, [vp, vp, i64, i64, i64, vp]),
    "mstts_transpose01": (i32, [vp, vp, i64, i64, i64, vp]),
    "mstts_speaker_finalize": (i32, [vp, vp, i64, i64, i64, i64, vp]),
    "mstts_adam_tf": (i32, [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, i64, vp]),
    "mstts_stft_mel": (i32, [vp, i64, f32, vp, vp, i32, i32, i32, i32, f32, vp, vp, i64, vp]),
    "mstts_stft_mel_ws_floats": (i64, [i64, i32, i64]),
    "mstts_stft_fft_supported": (i32, [i32, i32]),
    "mstts_stft_fft": (i32, [vp, vp, vp, i32, f32, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, i64, vp, vp, f32, i32, vp]),
    "mstts_fold_rows": (i32, [vp, vp, i64, i64, i64, i64, vp]),
    "mstts_decoder_bf16_splits": (i32, [i64, i64, i64, P(i32)]),
    "mstts_skinny_bf16_fwd_splits": (i32, [i64, i64]),
    "mstts_skinny_bf16_bwd_splits": (i32, [i64, i64]),
    "mstts_pack_bf16_fwd": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "mstts_pack_bf16_bwd": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "mstts_skinny_fwd_bf16": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_skinny_bwd_bf16": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_ge2e_ws_floats": (i64, [i64, i64, i64]),
    "mstts_ge2e_loss_fwd_bwd": (i32, [vp, i64, i64, i64, i64, vp, vp, vp, i64, vp, vp]),
    "mstts_wg_overlap_add": (i32, [vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "mstts_wg_gate": (i32, [vp, i64, vp, i64, i64, vp]),
    "mstts_wg_gate_add": (i32, [vp, i64, vp, vp, i64, i64, vp]),
    "mstts_wg_res_skip": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, vp]),
    "mstts_wg_coupling_inv": (i32, [vp, vp, vp, vp, f32, vp, i64, i64, i64, vp]),
    "mstts_philox_normal": (i32, [vp, i64, u64, u32, f32, vp]),
    "mstts_lstm_seq_fwd": (i32, [P(LstmSeqFwd), vp]),
    "mstts_lstm_seq_bwd": (i32, [P(LstmSeqBwd), vp]),
    "mstts_lstm_seq_fwd_pair": (i32, [P(LstmSeqFwd), P(LstmSeqFwd), vp]),
    "mstts_lstm_seq_bwd_pair": (i32, [P(LstmSeqBwd), P(LstmSeqBwd), vp]),
    "mstts_persist_lstm_supported": (i32, [i64, i64]),
    "mstts_persist_lstm_pack_floats": (i64, []),
    "mstts_persist_lstm_ws_bytes": (i64, []),
    "mstts_persist_lstm_pack": (i32, [vp, i64, vp, vp, vp]),
    "mstts_persist_lstm_supported_n": (i32, [i64, i64, i32]),
    "mstts_persist_lstm_fwd_supported_n": (i32, [i64, i64, i32]),
    "mstts_persist_lstm_pack_fwd": (i32, [vp, i64, i64, vp, vp]),
    "mstts_persist_lstm_ws_bytes_n": (i64, [i64, i32]),
    "mstts_persist_lstm_hist_floats_n": (i64, [i64, i64, i32]),
    "mstts_persist_lstm_bwd_floats_n": (i64, [i64, i64, i32]),
    "mstts_lstm_seq_fwd_persistent": (i32, [P(LstmSeqFwd), vp, vp, vp, vp, vp]),
    "mstts_lstm_seq_bwd_persistent": (i32, [P(LstmSeqBwd), vp, vp, vp, vp, vp, vp]),
    "mstts_persist_lstm_hist_floats": (i64, [i64]),
    "mstts_persist_lstm_bwd_floats": (i64, [i64]),
    "mstts_lstm_seq_fwd_pair_persistent": (i32, [P(LstmSeqFwd), P(LstmSeqFwd), vp, vp, vp, vp, vp, vp]),
    "mstts_lstm_seq_bwd_pair_persistent": (i32, [P(LstmSeqBwd), P(LstmSeqBwd), vp, vp, vp, vp, vp, vp, vp]),
    "mstts_lstm_point_bwd_pair": (i32, [P(LstmPointBwd), P(LstmPointBwd), vp]),
    "mstts_skinny_bwd_pair": (i32, [vp, vp, i64, vp, vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_lstm_seq_ws_floats": (i64, [i64, i64, i32]),
    "mstts_decoder_train_fwd": (i32, [P(DecoderTrain), vp]),
    "mstts_decoder_train_bwd": (i32, [P(DecoderTrainBwd), vp]),
    "mstts_persist_fwd_supported": (i32, [i64, i64, i64, i64, i64, i64]),
    "mstts_persist_fwd_ws_bytes": (i64, []),
    "mstts_persist_pack_floats": (i64, [i32]),
    "mstts_persist_pack": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "mstts_decoder_train_fwd_persistent": (i32, [P(DecoderTrain), P(PersistDesc), vp]),
    "mstts_persist_opk_floats": (i64, [i64]),
    "mstts_persist_unpack_history": (i32, [vp, P(DecoderTrain), vp]),
    "mstts_persist_bwd_supported": (i32, [i64, i64, i64, i64, i64, i64]),
    "mstts_persist_bwd_ws_bytes": (i64, []),
    "mstts_persist_bwd_pack_floats": (i64, [i32]),
    "mstts_persist_bwd_pack": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "mstts_decoder_train_bwd_persistent": (i32, [P(DecoderTrainBwd), P(PersistDesc), vp]),
    "mstts_decoder_train_bwd_ws_floats": (i64, [i64, i64, i64, i64, i64, i64]),
    "mstts_decoder_train_bwd_parts": (i32, [i64, i64]),
    "mstts_decoder_train_ws_floats": (i32, [i64, i64, i64, i64, C.POINTER(i64), C.POINTER(i64)]),
    "mstts_skinny_fwd_splits": (i32, [i64, i64]),
    "mstts_skinny_fwd": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_skinny_bwd_splits": (i32, [i64, i64]),
    "mstts_skinny_bwd": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_pack_skinny_bwd": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "mstts_skinny_bwd_packed": (i32, [vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "mstts_decoder_infer_fast": (i32, [i64, i64, i64, i64, i64, i64]),
    "mstts_decoder_infer_steps": (i32, [P(DecoderInfer), i64, i64, vp]),
    "mstts_persist_infer_supported": (i32, [i64, i64, i64, i64, i64, i64, i64, i64]),
    "mstts_persist_infer_ws_bytes": (i64, []),
    "mstts_persist_infer_pack_floats": (i64, []),
    "mstts_persist_infer_pack": (i32, [vp, vp, i64, vp, vp, vp]),
    "mstts_decoder_infer_persistent": (i32, [P(DecoderInfer), P(PersistInferDesc), vp]),
    "mstts_decoder_infer_ws_floats": (i64, [i64, i64, i64, i64, i64, i64]),
    "mstts_f32_to_bf16": (i32, [vp, vp, i64, vp]),
    "mstts_bf16_to_f32": (i32, [vp, vp, i64, vp]),
    "mstts_bf16_chunks_sum": (i32, [vp, i32, i64, i64, vp, vp]),
    "mstts_probe_begin": (i32, [i32, i64]),
    "mstts_probe_result": (i64, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


ABI_VERSION = 5          # = mstts_abi_version() of the library this binding's ctypes structs describe (bump both on a descriptor change)


class MsttsError(RuntimeError):
    pass


def load():
    """Load the HIP library, (re)building it first when its sources changed since it was built (multi_speaker_tts_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _b
    if _b.needs_build():                 # content hashes of the sources vs the ones the library was built from (file-locked)
        try:
            _b.build()
        except RuntimeError as e:
            # a deployed library WITHOUT its hash sidecar on a host without hipcc: load what is there (never a CPU fallback - without
            # the .so this still raises).  A sidecar that exists and disagrees with the sources means the library really is stale: its
            # descriptor structs may differ from the ctypes ones (symbols would still resolve), so that stays an error.
            if "hipcc not found" in str(e) and os.path.exists(LIB_PATH) and not os.path.exists(LIB_PATH + ".hash"):
                import warnings
                warnings.warn("libmstts_hip.so could not be checked against its sources (hipcc not found); loading the existing library")
            else:
                raise
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch; never fall back
        fn.restype = res
        fn.argtypes = args
    if lib.mstts_abi_version() != ABI_VERSION:     # descriptor layouts are not visible to the symbol check above
        raise MsttsError("libmstts_hip.so reports ABI version %d, this binding is written for %d: rebuild the library (python -m multi_speaker_tts_amd.build --force)"
                         % (lib.mstts_abi_version(), ABI_VERSION))
    # Development switches (A/B runs): the library itself reads no environment variable (include/mstts.h), this binding maps them onto
    # the process-global setters once, here.
    env = os.environ.get
    if env("MSTTS_GEMM_SPLIT3", "1") == "0":         # every fp32 contraction on the f32-input MFMA (see mstts_gemm_split3)
        lib.mstts_gemm_split3(0)
    if env("MSTTS_GEMM_SPLIT_BIG", "1") == "0":      # no 256 x 256-tile split kernel
        lib.mstts_gemm_split_big(0)
    if env("MSTTS_GEMM_BF16_BIG", "1") == "0":
        lib.mstts_gemm_bf16_big(0)
    if env("MSTTS_GEMM_BF16_AUTOCUT", "1") == "0":
        lib.mstts_gemm_bf16_autocut(0)
    if env("MSTTS_GEMM_SPLIT_BIG_MIN") or env("MSTTS_GEMM_BF16_BIG_MIN"):
        lib.mstts_gemm_big_min_workgroups(int(env("MSTTS_GEMM_SPLIT_BIG_MIN", "0")), int(env("MSTTS_GEMM_BF16_BIG_MIN", "0")))
    _lib = lib
    return lib


class deterministic_gemm:
    """Context manager / decorator: inside it (on this thread) mstts_gemm_f32 makes no K-cut of its own - one fixed summation order per output
    element, bit-reproducible run to run (mstts_gemm_deterministic).  Nests."""
    _depth = __import__("threading").local()

    def __enter__(self):
        n = getattr(self._depth, "n", 0)
        if n == 0:
            load().mstts_gemm_deterministic(1)
        self._depth.n = n + 1
        return self

    def __exit__(self, *exc):
        self._depth.n -= 1
        if self._depth.n == 0:
            load().mstts_gemm_deterministic(0)
        return False

    def __call__(self, fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            with deterministic_gemm():
                return fn(*a, **k)
        return wrapped


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # (what torch's own compiled code paths use: no Stream object per call)


def stream():
    """The current torch stream's handle for the C ABI.  A call through `call` costs 7.3 us of host time with `torch.cuda.current_stream()` (2.9 us of
    it building a Stream object) and 5 us with the raw getter - the inference forward's phases between the persistent launches are host-bound."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, offset=0):
    """Device pointer of a tensor (None -> NULL), advanced by ``offset`` elements."""
    if t is None:
        return None
    return t.data_ptr() + offset * t.element_size()


def call(name, *args):
    """Call an int-returning entry point on the current stream; raise MsttsError on failure."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise MsttsError("%s failed (%d): %s" % (name, rc, lib.mstts_last_error().decode()))


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, bias=None, trans_a=False, trans_b=False, act=ACT_NONE, accumulate=False,
         split_k=1, win=None, batch=1, strides=(0, 0, 0), alpha=1.0, a_off=0, b_off=0, c_off=0, bias_off=0, bf16=False):
    """Thin wrapper over mstts_gemm_f32 (bf16=True: mstts_gemm_bf16, operands rounded to bf16, fp32 accumulate).
    A/B/Cm/bias are tensors (used only for their pointers)."""
    d = GemmDesc()
    d.A, d.B, d.C, d.bias = ptr(A, a_off), ptr(B, b_off), ptr(Cm, c_off), ptr(bias, bias_off)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    if win is not None:
        d.win_T, d.win_C, d.win_pad = win[:3]
        d.win_dil = win[3] if len(win) > 3 else 1
    d.act, d.accumulate, d.split_k = act, int(accumulate), split_k
    d.batch, d.stride_a, d.stride_b, d.stride_c = batch, strides[0], strides[1], strides[2]
    d.alpha = alpha
    call("mstts_gemm_bf16" if bf16 else "mstts_gemm_f32", C.byref(d))
