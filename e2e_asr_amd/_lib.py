"""ctypes binding of libe2e_asr_hip.so (the C ABI declared in include/e2e_asr_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this
module raises, and every op in e2e_asr_amd.ops raises with it.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``e2e_asr_amd/csrc/build.sh``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASR_LIB_VARIANT=hunt: the debug build whose persistent kernels delay random publishers and pollers (csrc/common.h ASR_RACE_HUNT;
# tests/test_gpu_race_hunt.py) -- same ABI, never the product.  ASR_LIB_PATH: A/B builds of experiments.
_VARIANT = os.environ.get("ASR_LIB_VARIANT", "")
LIB_PATH = os.environ.get("ASR_LIB_PATH") or os.path.join(
    _HERE, "csrc", "libe2e_asr_hip_%s.so" % _VARIANT if _VARIANT else "libe2e_asr_hip.so")

c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p


class DecWeights(C.Structure):
    _fields_ = [(n, vp) for n in (
        "embedding", "attn_enc_w", "attn_v", "attn_w", "attn_b", "lm_kernel", "lm_bias",
        "dec_kernel", "dec_bias", "inp_w", "inp_b", "ap_w", "ap_b", "out_w", "out_b",
        "simple_w", "simple_b")]


class DecDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "Te", "D", "A", "H", "lmH", "E", "V", "T_out")]


class DecWs(C.Structure):
    _fields_ = [(n, vp) for n in (
        "hf", "tok", "lm_gates", "lm_c", "lm_h", "lm_hd", "sp", "x", "dec_gates", "dec_c",
        "dec_h", "alpha", "ctx", "p", "zeros", "y", "w2k", "chain_ws", "err",
        "lm_act", "lm_hprev", "lm_state", "lm_len", "lm_hx", "greedy_ws")]


class DecBwdWs(C.Structure):
    _fields_ = [(n, vp) for n in ("dP", "dQC", "dY", "dXH", "dLC", "dlm", "dEH", "dc_dec", "dc_lm", "dhf",
                                  "dv_part", "dctx", "emb_all", "chain_ws", "wc", "lm_hx")] + [("lm_deferred", C.c_int), ("side_busy", C.c_int)]


class LmWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("embedding", "lstm_kernel", "lstm_bias", "simple_w", "simple_b", "out_w", "out_b")] + \
               [(n, C.c_int) for n in ("E", "H", "P", "V")]


class BeamState(C.Structure):
    _fields_ = [(n, vp) for n in ("dc", "dh", "dlc", "dlh", "lc", "lh", "ctx")]


class BeamBook(C.Structure):
    _fields_ = [(n, vp) for n in ("ints", "cum", "state", "bp", "fin", "fin_score", "cand", "cand_idx")]


class LstmP3(C.Structure):
    """asr_lstm_p3 (include/e2e_asr_hip.h): plane operands of one (Bi)LSTM layer."""
    _fields_ = [("np", C.c_int), ("x_p3", vp), ("x_cols", C.c_int), ("kxT_p3", vp), ("out_p3", vp), ("hprev_p3", vp),
                ("dg_p3", vp), ("kxu_p3", vp), ("colmap", vp)]


class P3SplitJob(C.Structure):
    _fields_ = [("src", vp), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("dst", vp), ("np", C.c_int),
                ("transpose", C.c_int), ("dst_cols", C.c_int), ("unit_major_h", C.c_int)]


class DecGrads(C.Structure):
    _fields_ = [(n, vp) for n in (
        "embedding", "attn_enc_w", "attn_v", "attn_w", "attn_b", "lm_kernel", "lm_bias",
        "dec_kernel", "dec_bias", "inp_w", "inp_b", "ap_w", "ap_b", "out_w", "out_b",
        "simple_w", "simple_b")]


# name -> (restype, argtypes); every symbol include/e2e_asr_hip.h declares
SIGNATURES = {
    "asr_masked_ce_fwd_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "asr_p3_bytes": (C.c_size_t, [C.c_int] * 3),
    "asr_p3_split_multi": (C.c_int, [vp, C.c_int, C.POINTER(P3SplitJob)]),
    "asr_p3_split_ex": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_lstm_p3_supported": (C.c_int, [C.c_int] * 5),
    "asr_lstm_layer_fwd_p3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp,
                                        vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp, C.c_float, C.c_uint, vp, vp, C.POINTER(LstmP3)]),
    "asr_lstm_layer_bwd_p3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp,
                                        vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_float, C.c_uint, vp,
                                        C.POINTER(LstmP3)]),
    "asr_p3_split_f32": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    "asr_gemm_p3_rr": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int,
                                 vp, C.c_int, C.c_int, C.c_int, vp]),
    "asr_gemm_p3_kk": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int,
                                 vp, C.c_int, vp, C.c_int, C.c_int]),
    "asr_gemm_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp,
                               C.c_int, vp, C.c_int, vp, C.c_int]),
    "asr_gemm_f32_batched": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_longlong, vp,
                                       C.c_int, C.c_longlong, vp, C.c_int, C.c_longlong, vp, C.c_int, C.c_int]),
    "asr_lstm_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "asr_lstm_layer_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                     vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp,
                                     C.c_float, C.c_uint, vp, vp]),
    "asr_lstm_bwd_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "asr_lstm_layer_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                     vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                     vp, C.c_size_t, vp, C.c_float, C.c_uint, vp]),
    "asr_linear_wt_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_colsum_f32": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]),
    "asr_gather_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int]),
    "asr_concat2_multi": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "asr_scatter_add_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int]),
    "asr_sumsq_f32": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "asr_clip_adam_f32": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, vp] + [C.c_float] * 6),
    "asr_linear_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp,
                                 vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]),
    "asr_lstm_cell_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                    vp, vp, vp, vp, C.c_float, C.c_uint, C.c_uint]),
    "asr_attention_lds_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "asr_attention_fwd": (C.c_int, [vp, vp, C.c_int] + [vp] * 8 + [C.c_int] * 5),
    "asr_attention_shared_fwd": (C.c_int, [vp, vp, C.c_int] + [vp] * 8 + [C.c_int] * 6),
    "asr_masked_ce_fwd": (C.c_int, [vp] * 7 + [C.c_int] * 3),
    "asr_masked_ce_bwd": (C.c_int, [vp] * 7 + [C.c_int] * 3),
    "asr_next_token": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_uint, C.c_uint]),
    "asr_attn_decoder_bwd": (C.c_int, [vp, C.POINTER(DecWeights), C.POINTER(DecWeights), C.POINTER(DecDims),
                                       C.POINTER(DecWs), C.POINTER(DecBwdWs), vp, vp, vp, vp, C.c_float, C.c_uint]),
    "asr_scatter_add_rows_ld": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "asr_lstm_cell_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_uint, C.c_uint]),
    "asr_attn_cell_bwd": (C.c_int, [vp] * 11 + [vp, C.c_int] + [vp] * 6 + [vp, C.c_int, vp] + [C.c_int] * 5),
    "asr_side_join": (C.c_int, [vp]),
    "asr_side_wait": (C.c_int, [vp]),
    "asr_set_lstm_mfma": (C.c_int, [C.c_int]),
    "asr_get_lstm_mfma": (C.c_int, []),
    "asr_decoder_chain_supported": (C.c_int, [C.c_int] * 5),
    "asr_decoder_chain_rows": (C.c_int, [C.c_int]),
    "asr_decoder_chain_bwd_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_decoder_greedy_supported": (C.c_int, [C.c_int] * 8),
    "asr_decoder_greedy_ws_bytes": (C.c_size_t, [C.c_int] * 6),
    "asr_decoder_greedy_fwd": (C.c_int, [vp] * 22 + [C.c_int] * 9),
    "asr_pyramid_reduce_fwd": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 4),
    "asr_pyramid_reduce_bwd": (C.c_int, [vp, vp, vp] + [C.c_int] * 4),
    "asr_sigmoid_f32": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "asr_softmax_f32": (C.c_int, [vp, vp, vp, C.c_int]),
    "asr_beam_scratch_floats": (C.c_size_t, [C.c_int] * 5),
    "asr_beam_step": (C.c_int, [vp] * 13),
    "asr_beam_step_sel": (C.c_int, [vp] * 14),
    "asr_beam_step_perm": (C.c_int, [vp] * 17),
    "asr_lstm_kernel_tile_order": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "asr_beam_gather": (C.c_int, [vp, vp, C.c_int, vp, vp] + [C.c_int] * 4),
    "asr_beam_select": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]),
    "asr_beam_decode_ws_floats": (C.c_size_t, [vp, C.c_int, C.c_int]),
    "asr_beam_decode": (C.c_int, [vp] * 8 + [C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp, vp]),
    "asr_set_gemm_precision": (C.c_int, [C.c_int]),
    "asr_get_gemm_precision": (C.c_int, []),
    "asr_set_gemm_split": (C.c_int, [C.c_int]),
    "asr_get_gemm_split": (C.c_int, []),
    "asr_decoder_chain_ws_bytes": (C.c_size_t, [C.c_int] * 4),
    "asr_decoder_chain_bwd_ws_bytes": (C.c_size_t, [C.c_int] * 4),
    "asr_decoder_lm_chain_supported": (C.c_int, [C.c_int] * 2),
    "asr_zero_finished_rows": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "asr_resident_wg_budget": (C.c_int, []),
    "asr_gru_layer_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp,
                                    vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_uint, vp, vp]),
    "asr_gru_layer_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_uint, vp, vp]),
    "asr_attn_bwd": (C.c_int, [vp] * 11 + [C.c_int] + [vp] * 5 + [C.c_int] * 5),
    "asr_attn_decoder_bwd_lm": (C.c_int, [vp, C.POINTER(DecWeights), C.POINTER(DecWeights), C.POINTER(DecDims), C.POINTER(DecWs),
                                          C.POINTER(DecBwdWs), C.c_float, C.c_uint]),
    "asr_race_hunt_build": (C.c_int, []),
    "asr_set_wgrad_mode": (C.c_int, [C.c_int]),
    "asr_get_wgrad_mode": (C.c_int, []),
    "asr_scatter_add_rows_ordered": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int]),
    "asr_prof_enable": (C.c_int, [C.c_int]),
    "asr_prof_enable_mask": (C.c_int, [C.c_uint]),
    "asr_debug_set_buffer": (C.c_int, [vp]),
    "asr_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "asr_prof_read_each": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "asr_attn_decoder_fwd": (C.c_int, [vp, C.POINTER(DecWeights), C.POINTER(DecDims), C.POINTER(DecWs),
                                       vp, vp, vp, C.c_int, c_fp, C.c_float, C.c_float, C.c_uint, vp]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "e2e_asr_amd: HIP library not built (%s missing). Run __graft_entry__.build(). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if bool(lib.asr_race_hunt_build()) != (_VARIANT == "hunt") and not os.environ.get("ASR_LIB_PATH"):
        raise ImportError("e2e_asr_amd: %s is %sthe race-hunt debug build but ASR_LIB_VARIANT=%r" % (
            LIB_PATH, "" if lib.asr_race_hunt_build() else "not ", _VARIANT))
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
