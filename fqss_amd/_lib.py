"""ctypes binding of csrc/libfqss_hip.so (C ABI declared in include/fqss.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FQSS_LIB") or os.path.join(_HERE, "csrc", "libfqss_hip.so")   # FQSS_LIB: kernel A/B experiments

P, I64, I32, F32, F64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double

# name -> argtypes (restype is always int unless listed in _RESTYPE)
_PROTOS = {
    "fqss_version": [],
    "fqss_last_error": [],
    "fqss_selftest_div": [P, I64, F32, P, P],
    "fqss_actq_fwd": [P, P, P, I64, I64, I64, I64, I64, I32, P, I32, P, P, P, P],
    "fqss_obs_reset": [P, I64, P],
    "fqss_observer_ema": [P, P, P, F64, P],
    "fqss_actq_bwd": [P, P, P, I64, I64, I64, I64, I64, I32, P, I32, P, P, P, P, I64, P],
    "fqss_gluq_fwd": [P, P, I64, I64, I64, I64, I64, I32, P, P, P, P],
    "fqss_gluq_bwd": [P, P, P, I64, I64, I64, I64, I64, I64, I32, P, P, P, P],
    "fqss_actq_bwd_colbias": [P, P, P, I64, I32, I64, I64, I64, I32, P, I32, P, P, P, P, P],
    "fqss_actq2_bwd_colbias": [P, P, P, I64, I32, I64, I64, I64, P, P, P, P, P, P, P, P],
    "fqss_minmax": [P, I64, I64, I64, P, P],
    "fqss_wq_observe": [P, I64, I64, I64, P, P, P],
    "fqss_wq_fwd": [P, P, P, I64, I64, I64, P, P, P],
    "fqss_wq_bwd": [P, P, P, P, P, I64, I64, I64, P, P, I32, P],
    "fqss_gacc_flush": [P, P, P, P, P],
    "fqss_gacc_flush_multi": [P, I32, P],
    "fqss_wq_multi_fwd": [P, I32, I32, P],
    "fqss_wq_multi_bwd": [P, I32, I32, P],
    "fqss_pwconv_fwd": [P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_pwconv_fwd_x3": [P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_pwconv_fwd_x3s": [P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_pwconv_fwd_wq": [P, P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_conv1d_s1_fwd": [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_conv1d_s1_bwd_w": [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_halo_pack": [P, P, I64, I64, I64, I64, I64, I64, I64, I32, I32, I64, I64, P],
    "fqss_phase_pack": [P, P, I64, I64, I64, I64, I64, I64, I64, I32, I32, I32, I64, I64, P],
    "fqss_phase_unpack": [P, P, I64, I64, I64, I64, I64, I64, I64, I32, I32, I32, I64, I64, I64, I32, P, P],
    "fqss_conv2_fwd_wq": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_conv2_fwd_x3s": [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_conv2_bwd_x_wq": [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_conv2_bwd_w": [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_pwconv_bwd_x": [P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_pwconv_bwd_w": [P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_wq_codes": [P, P, P, P, P, I32, I32, P, P, P],
    "fqss_qpw_fwd": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_qpw_bwd_x": [P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_qpw_bwd_x_add": [P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_qpw_bwd_w": [P, P, P, P, P, I32, I32, I32, I32, I64, I64, P],
    "fqss_qpw_fwdq": [P, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, I64, I64, P, P],
    "fqss_qpw_fwdq_add": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, I64, I64, P, P, P],
    "fqss_add_chain_ok": [I64, I64, I32, I32],
    "fqss_add_chain_bwd": [P, I32, P, I64, P, I64, P, I64, P, I64, P, P, I64, I64, I32, I64, I64, I64, I64, P],
    "fqss_qpw_stat_slots": [I32, I32],
    "fqss_dwq_stat_slots": [I32, I32],
    "fqss_qpw_fwd2": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_qpw_bwd_x2": [P, P, P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_qpw_bwd_w2": [P, P, P, P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_qpw_bwd_w_group_ws": [P, I32],
    "fqss_qpw_bwd_w_group": [P, I32, P, I64, P],
    "fqss_decode": [P, P, I64, I64, I64, I64, P, P, P],
    "fqss_gnq_fwd": [P, P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, F32, P, P, P, P, I32, P],
    "fqss_gnq_bwd": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P, P, P, P, P],
    "fqss_ewq_bwd_p": [P, P, P, P, P, P, F32, P, P, I64, I64, I64, I64, I64, I64, I32, P, P, P, P, I32,
                       P, I64, I32, P, P, P, P, I64, P, I64, I32, P, P, P, P, I64, P],
    "fqss_gnq_bwd_p": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P, P, P, P, P, I64, I32, P, P, P, P],
    "fqss_dwq_fwd": [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, I64, I32, P, P, P, P, P],
    "fqss_dwq_bwd_z": [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, I64, I32, P, P, P, P, P, P],
    "fqss_ewq_fwd": [P, P, P, P, P, P, P, F32, P, P, I64, I64, I64, I64, I64, I64, I64, I32, P, P, P, P],
    "fqss_ewq_bwd": [P, P, P, P, P, P, P, F32, P, P, I64, I64, I64, I64, I64, I64, I64, I32, P, P, P, P, P],
    "fqss_dwq_bwd_w": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_mulq_fwd": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, I64, P, P, P],
    "fqss_mulq_bwd": [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, I64, I64, P, P, P, P, I64, I32, P, P, P, P],
    "fqss_dwq_bwd": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, I64, I32, P, P, P, P, P, P],
    "fqss_dwq_bwd_gn": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, I64, I32, P, P, P, P, P, P, P, P],
    "fqss_gnq_bwd_rows": [P, P, P, P, P, P, P, I32, I32, I32, I64, I64, P, P, P, P, P],
    "fqss_gnq_bwd_apply": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P, P, P, P, I64, I32, P, P, P, P],
    "fqss_split3_planes": [P, P, I64, P],
    "fqss_tgemm": [P, P, I32, I32, I32, I32, I64, I32, P, P, P, F32, P, P, I32, P, P, I32, P, P, I64, P, P, I64, P],
    "fqss_tgemm_tiled": [P, P, I32, I32, I32, I32, I64, I32, P, P, P, F32, P, P, I32, P, P, I32, P, P, I64, P, P, I64, P],
    "fqss_tgemm_tiled_ok": [I32, I32, I32],
    "fqss_split3_tiles": [P, P, I32, I32, P],
    "fqss_bn_moments": [P, P, I32, I32, I32, I64, P],
    "fqss_bn_apply": [P, P, P, P, I32, I32, I32, I64, I64, P],
    "fqss_bn_bwd_reduce": [P, P, P, I32, I32, I32, I64, I64, P],
    "fqss_bn_bwd_apply": [P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P],
    "fqss_tdw": [P, P, P, P, F32, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_tstats": [P, I32, I32, I32, I64, P, P],
    "fqss_dwconv_fwd": [P, P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_dwconv_bwd_x": [P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_dwconv_bwd_w": [P, P, P, I32, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_gn_fwd": [P, P, P, P, P, I32, I32, I32, I64, I64, F32, P, P],
    "fqss_gn_fwd_tail": [P, P, P, P, I32, I32, I32, I64, I64, F32, P, I32, P, P, I64, P],
    "fqss_gn_bwd": [P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P, P],
    "fqss_gnq_fwd_f": [P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, F32, P, P, P, P],
    "fqss_gnq_bwd_f": [P, P, P, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P, P, P, P, P],
    "fqss_axpby": [P, P, F32, F32, P, I64, I64, I64, I64, I64, P],
    "fqss_mul_bcast_fwd": [P, P, P, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_mul_bcast_bwd": [P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, I64, I64, P],
    "fqss_splitter2": [P, P, I32, I64, P, P],
    "fqss_splitter2_raw": [P, P, I32, I64, P, P],
    "fqss_frames_conv_fwd": [P, P, P, I32, I32, I32, I64, I32, I32, I32, I64, P],
    "fqss_frames_conv_add_fwd": [P, P, P, I64, P, I32, I32, I32, I64, I32, I32, I32, I64, P],
    "fqss_ola_convtr_fwd": [P, P, P, I32, I32, I32, I64, I32, I32, I64, P],
    "fqss_frames_wgrad1": [P, P, P, I32, I32, I32, I64, I64, I32, I32, P],
    "fqss_frames_wgrad1s": [P, P, I64, P, I64, I32, I32, I32, I64, I64, I32, I32, P],
    "fqss_frames_wgrad1_q": [P, P, P, P, P, I32, I32, I32, I64, I64, I32, I32, P],
    "fqss_ola_convtr_fwd_q": [P, P, P, P, P, I32, I32, I32, I64, I32, I32, I64, P],
    "fqss_ola_convtr_mul_fwd": [P, P, P, P, I32, I32, I32, I32, I64, I64, I32, I32, I64, P],
    "fqss_frames_wgrad": [P, P, P, I32, I32, I32, I32, I64, I64, I32, I32, P],
    "fqss_kd_loss": [P, P, P, I32, I64, F32, P, P, P, P, P, P],
    "fqss_kd_moments": [P, P, P, I32, I64, P, P],
    "fqss_pit_sisdr_loss": [P, P, I32, I64, P, P, P, P, P, P],
    "fqss_kd_loss_per_sample": [P, P, P, I32, I64, F32, I32, F32, P, P, P, P, P, P],
    "fqss_sumsq": [P, I64, P, P],
    "fqss_set_deterministic": [I32, P, I64, P],
    "fqss_det_finish": [P, I64, P, P],
    "fqss_adam_clip": [P, P, P, P, I64, P, F32, F32, F32, F32, F32, F32, P, P, P, P],
    "fqss_rowlin_fwd": [P, P, P, P, I64, I32, I32, I64, I64, I64, P],
    "fqss_rowlin_fwd_w3": [P, P, P, P, I64, I32, I32, I64, I64, P],
    "fqss_rowlin_bwd_x": [P, P, P, I64, I32, I32, I64, I64, I64, P],
    "fqss_rowlin_bwd_w": [P, P, P, I64, I32, I32, I64, I64, I64, P],
    "fqss_rowlin_bwd_w_batched": [P, P, P, I64, I32, I32, I64, I64, I64, I32, I64, I64, I64, P],
    "fqss_colsum": [P, P, I64, I32, I64, P],
    "fqss_layernorm_fwd": [P, P, P, P, P, I64, I32, I64, I64, F64, P],
    "fqss_layernorm_bwd": [P, P, P, P, P, P, P, I64, I32, I64, I64, I64, P],
    "fqss_layernormq_fwd": [P, P, P, P, P, P, I64, I32, I64, I64, I64, F64, P, P, P],
    "fqss_layernormq_bwd": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, P, P, P, P],
    "fqss_unary_fwd": [P, P, I64, I32, F64, P],
    "fqss_unary_rows_fwd": [P, P, I64, I32, I64, I64, I32, F64, P],
    "fqss_unary2_fwd": [P, P, I64, I32, F64, F64, P],
    "fqss_unary_bwd": [P, P, P, I64, I32, F64, P],
    "fqss_permute4": [P, P, I64, I64, I64, I32, I64, I64, I64, P],
    "fqss_permute4_ld": [P, P, I64, I64, I64, I32, I64, I64, I64, I64, P],
    "fqss_dp_segment_fwd": [P, P, I32, I32, I64, I64, I32, I32, P],
    "fqss_dp_segment_bwd": [P, P, I32, I32, I64, I64, I32, I32, P],
    "fqss_dp_merge_fwd": [P, P, P, I32, I32, I32, I32, I32, I64, I64, P],
    "fqss_dp_merge_bwd": [P, P, P, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_ola2_fwd": [P, P, I64, I64, I64, P],
    "fqss_ola2_bwd": [P, P, I64, I64, I64, P],
    "fqss_attn_fwd": [P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, I64, P, P, P],
    "fqss_attn_bwd": [P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "fqss_add_layernorm_fwd": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, I64, I64, F64, P, P, P],
    "fqss_add_layernorm_bwd": [P, P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, I64, P, P, P, P],
    "fqss_addq_layernorm_fwd": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, I64, I64, F64, P, P, P, P, P],
    "fqss_addq_layernorm_bwd": [P, P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, I64, P, P, P, P, P, P, P],
    "fqss_addq_layernorm_fwd_map": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, F64, P, P, P, P, I64, I64, I64, I64, I64, P],
    "fqss_addq_layernorm_bwd_map": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, P, P, P, P, P, P, I64, I64, I64, I64, I64, P],
    "fqss_mha_prep_fwd": [P, P, P, P, I64, I32, I64, F64, P, P],
    "fqss_mha_prep_fwd_c": [P, P, P, P, I64, I32, I64, F64, P, P],
    "fqss_mha_prep_bwd": [P, P, P, P, P, I64, I32, I64, I64, F64, P, P, P],
    "fqss_lstm_fwd": [P, P, P, P, P, P, I32, I32, I32, P],
    "fqss_lstm_bwd": [P, P, P, P, P, I32, I32, I32, P],
    "fqss_lstm_gate_fn": [P, P, P, I64, P],
    "fqss_lstm_bwd_b": [P, P, P, P, P, P, I32, I32, I32, P],
    "fqss_lstm_bwd_b4": [P, P, P, P, P, P, I32, I32, I32, P],
    "fqss_gnrows_fwd": [P, P, P, P, P, P, I64, I32, I64, I64, I32, I32, I32, F64, P],
    "fqss_gnrows_bwd": [P, P, P, P, P, P, P, P, I64, I32, I64, I64, I64, I32, I32, I32, P],
    "fqss_bcast_add": [P, P, P, I64, I64, I32, P],
    "fqss_bcast_sum": [P, P, I64, I64, I32, P],
    "fqss_qrow_fwd": [P, P, P, P, P, P, P, P, I64, I32, I32, I64, I64, P],
    "fqss_qrow_fwdq": [P, P, P, P, P, P, P, P, P, I64, I32, I32, I64, I64, I64, I32, P, P, P, P],
    "fqss_qrow_fwdq2": [P, P, P, P, P, P, P, P, P, P, I64, I32, I32, I64, I64, I64, I64, P, P, P, P, P],
    "fqss_qrow_bwd_x": [P, P, P, P, I64, I32, I32, I64, I64, P],
    "fqss_qrow_bwd_w": [P, P, P, P, P, I64, I32, I32, I64, I64, I64, P],
    "fqss_qrow_bwd_wb": [P, P, P, P, P, P, I64, I32, I32, I64, I64, I64, P],
    "fqss_qrow_bwd_w_batched": [P, P, P, P, P, I64, I32, I32, I64, I64, I64, I32, I64, I64, I64, P],
    "fqss_qrow_bwd_w_group": [P, I32, P],
    "fqss_glu_fwd": [P, P, I64, I64, I64, I64, I64, P],
    "fqss_glu_bwd": [P, P, P, I64, I64, I64, I64, I64, I64, P],
    "fqss_div_fwd": [P, P, P, I64, P],
    "fqss_div_bwd": [P, P, P, P, P, I64, P],
    "fqss_embedding_fwd": [P, P, P, I64, I32, I64, P],
    "fqss_embedding_bwd": [P, P, P, I64, I32, I64, P],
    "fqss_frames_gather": [P, P, I64, I64, I64, I64, I64, I64, I64, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_frames_ola": [P, P, P, I64, I64, I64, I64, I64, I64, I64, I32, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, P],
    "fqss_chan_sum": [P, P, I64, I64, I64, I64, P],
    "fqss_chan_op": [P, P, P, I64, I64, I64, I64, I64, I32, P],
    "fqss_chan_scale_bwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "fqss_col_scale_fwd": [P, P, P, I64, I32, I64, I64, P],
    "fqss_col_scale_bwd": [P, P, P, P, P, I64, I32, I64, I64, I64, P],
    "fqss_sample_meanstd": [P, P, P, I64, I64, P],
    "fqss_sample_norm": [P, P, P, I64, I64, I32, P],
    "fqss_stft": [P, P, P, P, I64, I64, I64, I32, I32, I32, I32, P],
    "fqss_istft": [P, P, P, P, P, P, I64, I64, I64, I32, I32, I32, I32, P],
    "fqss_istft_bwd": [P, P, P, P, P, I64, I64, I64, I32, I32, I32, I32, P],
    "fqss_transpose2d": [P, P, I64, I64, I64, P],
    "fqss_hd_kd_loss": [P, P, P, P, P, P, P, P, I32, I32, I64, F32, P],
    "fqss_sisnr_matrix": [P, P, P, P, P, I32, I64, I64, I64, P],
    "fqss_infer_ola": [P, P, P, P, I32, I32, I64, I64, I64, I64, I64, P],
    "fqss_infer_normalize": [P, P, I64, I64, I64, P],
    "fqss_fq_affine": [P, P, P, I64, I64, I64, P, P, I32, I32, P],
    "fqss_snr_mix": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, I32, P],
    "fqss_resample_fir": [P, P, P, I64, I64, I64, I64, I64, I32, I32, I32, P],
    "fqss_attn_long_fwd": [P, P, P, P, P, I32, I32, I32, I32, I32, P, P, P, P],
    "fqss_attn_long_bwd": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P],
    "fqss_attn_long_fwd_c": [P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P],
    "fqss_attn_long_bwd_c": [P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, P],
    # descriptor-struct forms (csrc/desc_api.hip); the structs are below
    "fqss_workspace_bytes": [C.c_char_p, P, I32],
    "fqss_add_fq_fwd": [P, P, P, P, F32, P, P, P, P, C.c_size_t, P],
    "fqss_add_fq_bwd": [P, P, P, P, F32, P, P, P, P, P, P, C.c_size_t, P],
    "fqss_pwconv_fq_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, C.c_size_t, P],
    "fqss_gln_fq_fwd": [P, P, P, P, F32, P, P, P, P, P, C.c_size_t, P, I32, P],
    "fqss_tgemm_desc": [P, P],
}
_RESTYPE = {"fqss_last_error": C.c_char_p, "fqss_workspace_bytes": C.c_int64, "fqss_qpw_bwd_w_group_ws": C.c_int64}

DT_F32, DT_U8, DT_I8, DT_F64, DT_U16, DT_I64 = range(6)


class FqssTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int), ("ndim", C.c_int), ("shape", C.c_int64 * 4), ("stride", C.c_int64 * 4)]


class FqssQParams(C.Structure):
    _fields_ = [("qmin", C.c_void_p), ("qmax", C.c_void_p), ("act", C.c_int), ("slope", C.c_void_p), ("gacc", C.c_void_p)]


class FqssWCodes(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("idxT", C.c_void_p), ("dw", C.c_void_p), ("rw", C.c_void_p), ("Co", C.c_int), ("Ci", C.c_int)]


class FqssAddChainLevel(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ac", "bc", "bz", "bout", "amin", "amax", "bmin", "bmax", "qmin", "qmax", "gacc", "bgacc", "bgbias")]


class FqssAddAfter(C.Structure):
    _fields_ = [("a", C.c_void_p), ("ld_a", C.c_int64), ("amin", C.c_void_p), ("amax", C.c_void_p), ("qmin", C.c_void_p),
                ("qmax", C.c_void_p), ("y", C.c_void_p), ("ld_y", C.c_int64)]


class FqssGnAfter(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean_rstd", C.c_void_p), ("ws", C.c_void_p), ("qmin", C.c_void_p),
                ("qmax", C.c_void_p), ("ggamma", C.c_void_p), ("gbeta", C.c_void_p)]


class FqssGnBefore(C.Structure):
    _fields_ = [("xc0", C.c_void_p), ("ld_xc0", C.c_int64), ("qmin0", C.c_void_p), ("qmax0", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("mean_rstd", C.c_void_p), ("ws", C.c_void_p), ("gacc", C.c_void_p)]


class FqssRowWgradJob(C.Structure):
    _fields_ = [("gz", C.c_void_p), ("xc", C.c_void_p), ("qmin_x", C.c_void_p), ("qmax_x", C.c_void_p), ("gw", C.c_void_p),
                ("gbias", C.c_void_p), ("R", C.c_int64), ("Ci", C.c_int32), ("Co", C.c_int32), ("ld_gz", C.c_int64),
                ("ld_xc", C.c_int64), ("ld_gw", C.c_int64)]


class FqssWgradJob(C.Structure):
    _fields_ = [("gz1", C.c_void_p), ("gz2", C.c_void_p), ("xc", C.c_void_p), ("qmin_x", C.c_void_p), ("qmax_x", C.c_void_p),
                ("gw", C.c_void_p), ("B", C.c_int32), ("Ci", C.c_int32), ("Co1", C.c_int32), ("Co2", C.c_int32), ("M", C.c_int32),
                ("ld_gz1", C.c_int64), ("ld_gz2", C.c_int64), ("ld_xc", C.c_int64)]


class FqssProducer(C.Structure):
    _fields_ = [("z", C.POINTER(FqssTensor)), ("act", C.c_int), ("slope", C.c_void_p), ("gacc", C.c_void_p), ("gbias", C.c_void_p),
                ("out", C.POINTER(FqssTensor))]


class FqssTGemmDesc(C.Structure):
    _fields_ = [("planes", C.POINTER(FqssTensor)), ("x", C.POINTER(FqssTensor)), ("pro", C.c_int), ("pro_stats", C.c_void_p),
                ("pro_gamma", C.c_void_p), ("pro_beta", C.c_void_p), ("pro_eps", C.c_float), ("pro_slope", C.c_void_p),
                ("bias", C.c_void_p), ("act", C.c_int), ("slope", C.c_void_p), ("stats_out", C.c_void_p), ("M1", C.c_int),
                ("c1", C.POINTER(FqssTensor)), ("r1", C.POINTER(FqssTensor)), ("c2", C.POINTER(FqssTensor)),
                ("r2", C.POINTER(FqssTensor))]

EXPORTS = tuple(_PROTOS)
# entry points of include/fqss_experiments.h: only in variants/libfqss_experiments.so (`make -C fqss_amd/csrc experiments`, FQSS_LIB)
_PROTOS.update({
    "fqss_gndwq_fwd": [P, P, P, P, P, F32, P, I32, P, P, P, P, P, P, I32, I32, I32, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P],
    "fqss_gndwq_fwd_v1": [P, P, P, P, P, F32, P, I32, P, P, P, P, P, P, I32, I32, I32, P, P, P, P, P, I32, I32, I32, I64, I64, I64, P],
})


class FqssError(RuntimeError):
    pass


_lib = None
BACKEND = "hip"           # "hip": csrc/libfqss_hip.so on MI355X (the product) | "cpu": csrc/cpu/libfqss_cpu.so (cfg 1: `--use_cpu`)
CPU_SO_PATH = os.path.join(_HERE, "csrc", "cpu", "libfqss_cpu.so")
_cpu = None


def set_backend(name):
    """`--use_cpu` (reference train.py:31) selects the CPU backend behind the same C ABI: a build-owned plain-C++ library that serves the
    entry points of the un-fused ConvTasNet QAT step (BASELINE.json configs[0]).  It is chosen EXPLICITLY, never as a fallback: with the
    HIP backend selected a CPU tensor or a missing .so still raises."""
    global BACKEND
    if name not in ("hip", "cpu"):
        raise FqssError(f"unknown backend {name!r}")
    if name == "cpu":
        load_cpu()
    BACKEND = name
    _bound.clear()
    from . import ops
    ops.CODED = name != "cpu"        # the CPU backend has the fp32 per-layer kernels only: no layer output carries codes there


def load_cpu():
    global _cpu
    if _cpu is None:
        if not os.path.exists(CPU_SO_PATH):
            raise FqssError(f"{CPU_SO_PATH} not found: build it with `make -C fqss_amd/csrc/cpu` (or __graft_entry__.build())")
        _cpu = C.CDLL(CPU_SO_PATH)
        _cpu.fqss_last_error.restype = C.c_char_p
    return _cpu


def load(strict=False):
    """Load the HIP library; fail LOUDLY when it is missing (no CPU fallback exists).
    strict=True additionally requires every symbol declared in include/fqss.h to be exported."""
    global _lib
    if BACKEND == "cpu" and not strict:
        return load_cpu()
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise FqssError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C fqss_amd/csrc`). fqss_amd has no CPU fallback.")
        _lib = C.CDLL(SO_PATH)
        _lib.fqss_last_error.restype = C.c_char_p
    if strict:
        missing = [n for n in EXPORTS if not hasattr(_lib, n)]
        if missing:
            raise FqssError(f"{SO_PATH} does not export: {missing}")
    return _lib


_bound = {}


def _bind(name):
    fn = _bound.get(name)
    if fn is None:
        lib = load()
        if not hasattr(lib, name):
            raise FqssError(f"{name} is not built for the {BACKEND} backend" + (
                " (the CPU backend serves the un-fused ConvTasNet QAT step only: cfg 1 of BASELINE.json)" if BACKEND == "cpu" else ""))
        fn = getattr(lib, name)
        fn.argtypes = _PROTOS[name]
        fn.restype = _RESTYPE.get(name, C.c_int)
        _bound[name] = fn
    return fn


def call(name, *args):
    rc = _bind(name)(*args)
    if rc != 0:
        raise FqssError(f"{name} failed ({rc}): {load().fqss_last_error().decode()}")


def query(name, *args):
    """entry points that return a count instead of a status (fqss_*_stat_slots, fqss_workspace_bytes)"""
    return _bind(name)(*args)
