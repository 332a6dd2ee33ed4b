"""ctypes binding of libds_kernels.so (the C ABI declared in include/ds_kernels.h).

The library is the product: there is NO CPU or PyTorch fallback.  If the shared object is
missing or a call fails, a RuntimeError is raised immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DS_LIB: an alternatively built library (kernel variants measured side by side on one box; tuning aid)
LIB_PATH = os.environ.get("DS_LIB") or os.path.join(_HERE, "libds_kernels.so")

def tuning_env(name, default=None):
    """The Python engine's A/B switches (DS_ZCAT, DS_FUSE_B3, DS_STEM_POOL, ...): like the C library's knobs they are honoured
    only beside the tuning build (DS_LIB set, scripts/_tuning.py); the product path reads none of them and runs the defaults."""
    return os.environ.get(name, default) if os.environ.get("DS_LIB") else default


DS_EPI_BIAS, DS_EPI_RELU, DS_EPI_ACCUM, DS_EPI_STATS, DS_EPI_MASK, DS_EPI_BNSUMS = 1, 2, 4, 8, 16, 32
DS_DTYPE_F32, DS_DTYPE_BF16 = 0, 1
DS_FP8_E4M3, DS_FP8_E5M2 = 0, 1
DS_CONV_FWD, DS_CONV_DGRAD = 0, 1
DS_LSTM_SKIP_MASKED = 256
DS_ARITH_F32, DS_ARITH_BF16, DS_ARITH_FP8, DS_ARITH_F32X3 = 0, 1, 2, 3
DS_FAM_IGEMM, DS_FAM_WINO2, DS_FAM_WINO4, DS_FAM_STEM, DS_FAM_BF16D, DS_FAM_FP8D, DS_FAM_F32X3, DS_FAM_WINO4H, DS_FAM_STEM_POOL = range(9)
DS_PLAN_NO_WINO, DS_PLAN_NO_WINO4, DS_PLAN_NO_STEM_DIRECT, DS_PLAN_NO_BF16_DIRECT, DS_PLAN_ACT16, DS_PLAN_PACKED_RGB = 1, 2, 4, 8, 16, 32
DS_PLAN_FP8_EVERYWHERE, DS_PLAN_FP8_WIDE_RULE, DS_PLAN_NO_WINO4H, DS_PLAN_STEM_POOL = 64, 128, 256, 512
DS_PLAN_NO_SPLITK = 1024


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("ldz", C.c_int32),
        ("w_tap_stride", C.c_int64), ("w_n_stride", C.c_int32), ("w_k_stride", C.c_int32),
        ("flip", C.c_int32), ("fold_cin", C.c_int32), ("flags", C.c_int32), ("ldmask", C.c_int32),
        ("splits", C.c_int32), ("z_split_stride", C.c_int64),
        ("tile_nt", C.c_int32), ("grid_x", C.c_int32), ("dtype", C.c_int32), ("x_dtype", C.c_int32),
        ("partials", C.c_int32),
        ("norm_rstd", C.c_void_p), ("norm_shift", C.c_void_p), ("mask_rstd", C.c_void_p), ("mask_shift", C.c_void_p),
        ("bnb", C.c_void_p), ("mask_dtype", C.c_int32), ("z_dtype", C.c_int32), ("pool_argmax", C.c_void_p), ("fin", C.c_void_p),
    ]


class BnBwdOnLoad(C.Structure):
    """ds_bn_bwd_on_load"""
    _fields_ = [("mean", C.c_void_p), ("rstd", C.c_void_p), ("shift", C.c_void_p), ("coef", C.c_void_p),
                ("nseg", C.c_int32), ("c_end", C.c_int32 * 3), ("ld", C.c_int32 * 3), ("dy", C.c_void_p * 3)]


class BnFinalizeInLaunch(C.Structure):
    """ds_bn_finalize_in_launch"""
    _fields_ = [("beta", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("shift", C.c_void_p),
                ("moving_mean", C.c_void_p), ("moving_var", C.c_void_p), ("ticket", C.c_void_p), ("count", C.c_int64),
                ("eps", C.c_float), ("decay", C.c_float)]


class LayerPlanStruct(C.Structure):
    """ds_conv_layer_plan"""
    _fields_ = [("d", ConvDesc), ("family", C.c_int32), ("role", C.c_int32), ("arith", C.c_int32), ("partials", C.c_int32),
                ("w_cin", C.c_int32), ("w_cout", C.c_int32), ("k", C.c_int32), ("a_format", C.c_int32), ("x16_ok", C.c_int32),
                ("w_bytes", C.c_int64), ("wscale_floats", C.c_int64), ("alg_flops", C.c_double), ("splitk", C.c_int32),
                ("reserved0", C.c_int32), ("ws_bytes", C.c_int64)]


class ConvIO(C.Structure):
    """ds_conv_io"""
    _fields_ = [("bias", C.c_void_p), ("mask", C.c_void_p), ("stats", C.c_void_p), ("pivot", C.c_void_p),
                ("x_amax", C.c_void_p), ("wscale", C.c_void_p), ("fin", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class Segments(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("c_begin", C.c_int32 * 4), ("c_end", C.c_int32 * 4),
                ("ld", C.c_int32 * 4), ("ptr", C.c_void_p * 4), ("dtype", C.c_int32 * 4),
                ("amax", C.c_void_p * 4), ("ptr2", C.c_void_p * 4)]


class BnFinalizeJob(C.Structure):
    """ds_bn_finalize_job"""
    _fields_ = [("stats", C.c_void_p), ("P", C.c_int32), ("C", C.c_int32), ("count", C.c_int64), ("beta", C.c_void_p),
                ("pivot", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("shift", C.c_void_p),
                ("moving_mean", C.c_void_p), ("moving_var", C.c_void_p)]


class SumSegments(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("c_begin", C.c_int32 * 4), ("c_end", C.c_int32 * 4), ("P", C.c_int32 * 4),
                ("kind", C.c_int32 * 4), ("s", C.c_void_p * 4), ("q", C.c_void_p * 4), ("P2", C.c_int32 * 4),
                ("s2", C.c_void_p * 4), ("q2", C.c_void_p * 4)]


_P = C.c_void_p
_i32, _i64, _f32, _u64 = C.c_int32, C.c_int64, C.c_float, C.c_uint64
_CD = C.POINTER(ConvDesc)
_SG = C.POINTER(Segments)
_SS = C.POINTER(SumSegments)
_LP = C.POINTER(LayerPlanStruct)
_IO = C.POINTER(ConvIO)

# name -> (restype, argtypes); mirrors include/ds_kernels.h one to one
SIGNATURES = {
    "ds_version": (C.c_int, []),
    "ds_last_error": (C.c_char_p, []),
    "ds_conv_igemm_partials": (C.c_int, [_CD]),
    "ds_conv_igemm_bnsums_supported": (C.c_int, [_CD]),
    "ds_conv_igemm_norm_supported": (C.c_int, [_CD]),
    "ds_conv_igemm_bnb_supported": (C.c_int, [_CD]),
    "ds_conv_igemm_pool3_supported": (C.c_int, [_CD]),
    "ds_conv_igemm_finalize_tickets": (C.c_int, [_CD]),
    "ds_conv_igemm": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ds_weights_bf16_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "ds_weights_to_bf16": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_bf16_supported": (C.c_int, [_CD]),
    "ds_conv_bf16_partials": (C.c_int, [_CD]),
    "ds_conv_bf16": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _P]),
    "ds_weights_f32x3_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "ds_weights_to_f32x3": (C.c_int, [_P, _P, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_f32x3_supported": (C.c_int, [_CD]),
    "ds_conv_f32x3_partials": (C.c_int, [_CD]),
    "ds_conv_f32x3": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _P]),
    "ds_absmax": (C.c_int, [_P, _i64, _i32, _P, _P]),
    "ds_weights_fp8_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "ds_weights_to_fp8": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_fp8_supported": (C.c_int, [_CD]),
    "ds_conv_fp8_partials": (C.c_int, [_CD]),
    "ds_conv_fp8": (C.c_int, [_CD, _P, _P, _i32, _P, _P, _P, _P, _P, _P, _P]),
    "ds_conv_stem_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_stem_bf16_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_stem": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_stem_bf16": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_stem_pool_supported": (C.c_int, [_i32, _i32]),
    "ds_conv_stem_pool_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_stem_pool": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_stem_pool_bf16": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_wino_transform_weights": (C.c_int, [_P, _P, _i32, _i32, _i32, _P]),
    "ds_conv_wino_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_wino": (C.c_int, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_wino4_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "ds_conv_wino4_prefer": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "ds_wino4_transform_weights": (C.c_int, [_P, _P, _i32, _i32, _i32, _P]),
    "ds_conv_wino4_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_wino4_splitk_choose": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "ds_conv_wino4_splitk_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_conv_wino4_splitk_workspace": (C.c_size_t, [_i32, _i32, _i32, _i32, _i32]),
    "ds_conv_wino4_splitk": (C.c_int, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P, C.c_size_t, _P]),
    "ds_conv_wino4": (C.c_int, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_wino4_bf16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_conv_wino4_bf16x2_x16": (C.c_int, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "ds_wino4_transform_weights_bf16x2": (C.c_int, [_P, _P, _i32, _i32, _i32, _P]),
    "ds_conv_plan": (C.c_int, [_LP, _i32, _i32, C.c_uint32] + [_i32] * 10),
    "ds_conv_plan_set_flags": (C.c_int, [_LP, _i32]),
    "ds_conv_plan_enable_bnsums": (C.c_int, [_LP, _i32]),
    "ds_conv_plan_norm_supported": (C.c_int, [_LP]),
    "ds_conv_plan_bnb_supported": (C.c_int, [_LP]),
    "ds_conv_plan_enable_pool3": (C.c_int, [_LP, C.c_void_p]),
    "ds_conv_plan_finalize_tickets": (C.c_int, [_LP]),
    "ds_conv_prepare_weights": (C.c_int, [_LP, _P, _P, _P, _P]),
    "ds_conv_run": (C.c_int, [_LP, _P, _P, _P, _IO, _P]),
    "ds_conv_wgrad_workspace": (C.c_size_t, [_CD]),
    "ds_conv_wgrad": (C.c_int, [_CD, _P, _P, _i32, _P, _P, C.c_size_t, _P]),
    "ds_bn_finalize": (C.c_int, [_P, _i32, _i64, _i32, _P, _P, _f32, _f32, _P, _P, _P, _P, _P, _P]),
    "ds_bn_finalize_centered": (C.c_int, [_P, _i32, _i64, _i32, _P, _P, _f32, _f32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ds_bn_apply_relu": (C.c_int, [_P, _i64, _i32, _P, _P, _SG, _P]),
    "ds_bn_apply_relu_z16": (C.c_int, [_P, _i64, _i32, _P, _P, _SG, _P]),
    "ds_bn_bwd_apply_z16": (C.c_int, [_P, _i32, _SG, _i64, _i32, _P, _P, _P, _P, _P, _i32, _P, _P]),
    "ds_bn_infer_prepare": (C.c_int, [_P, _P, _P, _f32, _i32, _P, _P, _P]),
    "ds_bn_bwd_partials": (C.c_int, [_i64, _i32]),
    "ds_bn_bwd_reduce": (C.c_int, [_P, _i32, _i32, _SG, _i64, _i32, _P, _P, _P, _P, _P]),
    "ds_bn_bwd_finalize_segs": (C.c_int, [_SS, _i64, _i32, _P, _P, _P, _P]),
    "ds_bn_bwd_finalize": (C.c_int, [_P, _i32, _i64, _i32, _P, _P, _P]),
    "ds_bn_bwd_finalize_multi": (C.c_int, [_SS, _i64, _i32, _P, _P, _P, _P]),
    "ds_bn_finalize_multi": (C.c_int, [_P, _i32, _f32, _f32, _P]),
    "ds_bn_finalize_apply_relu": (C.c_int, [_P, _i32, _i64, _i32, _P, _P, _f32, _f32, _P, _P, _P, _P, _P, _P, _i64, _SG, _P, _P]),
    "ds_bn_bwd_finalize_apply": (C.c_int, [_SS, _P, _P, _P, _P, _P, _P, _i32, _SG, _i64, _i32, _P, _P, _P, _P, _i32, _i32, _P, _P, _P]),
    "ds_bn_bwd_apply": (C.c_int, [_P, _i32, _SG, _i64, _i32, _P, _P, _P, _P, _P, _P, _P]),
    "ds_bn_bwd_apply_bf16": (C.c_int, [_P, _i32, _SG, _i64, _i32, _P, _P, _P, _P, _P, _i32, _P, _P]),
    "ds_maxpool_fwd": (C.c_int, [_P, _P, _P] + [_i32] * 11 + [_P]),
    "ds_maxpool_bn_relu_fwd": (C.c_int, [_P, _P, _P, _P, _P] + [_i32] * 11 + [_P, _P]),
    "ds_bn_pool_bwd_partials": (C.c_int, [_i32, _i32, _i32, _i32]),
    "ds_bn_pool_bwd_reduce": (C.c_int, [_P, _P, _P] + [_i32] * 8 + [_P, _P, _P, _P, _P]),
    "ds_bn_pool_bwd_apply": (C.c_int, [_P, _P, _P] + [_i32] * 8 + [_P, _P, _P, _P, _P, _P]),
    "ds_maxpool_bwd": (C.c_int, [_P, _P, _P] + [_i32] * 11 + [_P]),
    "ds_maxpool3_bwd_sums_partials": (C.c_int, [_i32, _i32, _i32]),
    "ds_maxpool3_bwd_sums": (C.c_int, [_P, _P, _P, _i32, _P, _i32, _i32, _i32, _i32, _i32, _P, _P]),
    "ds_maxpool3_bwd_dy16": (C.c_int, [_P, _P, _P, _i32, _P, _i32, _i32, _i32, _i32, _i32, _P, _P]),
    "ds_avgpool_dropout_fwd": (C.c_int, [_P, _i32, _i32, _i32, _f32, _u64, _P, _P, _P, _P, _P]),
    "ds_avgpool_dropout_bwd": (C.c_int, [_P, _P, _i32, _i32, _i32, _f32, _P, _P]),
    "ds_gather_rows": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i64, _i32, _P]),
    "ds_embedding_grad": (C.c_int, [_P, _P, _P, _i32, _i32, _i32, _i64, _i32, _P]),
    "ds_lstm_cell_fwd": (C.c_int, [_P, _P, _i32, _i64, _P, _P, _P, _i32, _i32, _i32, _f32, _P, _P, _P]),
    "ds_lstm_cell_bwd": (C.c_int, [_P, _P, _P, _P, _P, _i32, _i64, _P, _P, _i32, _i32, _i32, _P, _P, _P, _P]),
    "ds_seq_sort_desc": (C.c_int, [_P, _i32, _i32, _P, _P, _P]),
    "ds_permute_rows": (C.c_int, [_P, _i64, _P, _i64, _P, _i32, _i32, _i32, _i32, _P]),
    "ds_lstm_seq_supported": (C.c_int, [_i32, _i32]),
    "ds_lstm_seq_workspace": (C.c_size_t, [_i32, _i32]),
    "ds_lstm_seq_fwd": (C.c_int, [_P, _P, _i32, _P, _P, _P, _i32, _i32, _i32, _f32, _i32, _P, C.c_size_t, _P]),
    "ds_lstm_seq_bwd": (C.c_int, [_P, _P, _i32, _P, _P, _i32, _P, _i32, _i32, _i32, _P, _i32, _P, C.c_size_t, _P]),
    "ds_lstm_seq_status": (C.c_int, [_P, _i32]),
    "ds_softmax_ce": (C.c_int, [_P, _P, _i32, _i32, _f32, _P, _P, _P, _P]),
    "ds_adam_tf": (C.c_int, [_P, _P, _P, _P, _i64, _i64, _f32, _f32, _f32, _P, _f32, _f32, _f32, _P]),
    "ds_sumsq": (C.c_int, [_P, _i64, _P, _P, _P]),
    "ds_colsum": (C.c_int, [_P, _i64, _i32, _i32, _P, _P, _P]),
    "ds_slab_epilogue": (C.c_int, [_P, _i32, _i64, _i32, _i64, _i32, _P, _i32, _P, _P, _i32, _i32, _P]),
    "ds_copy2d": (C.c_int, [_P, _i32, _P, _i32, _i64, _i32, _P]),
    "ds_pad_channels": (C.c_int, [_P, _i32, _P, _i32, _i64, _P]),
    "ds_fill": (C.c_int, [_P, _i64, _f32, _P]),
}

_lib = None


# the -DDS_TUNING build only (libds_kernels_tuning.so): process-global switches for kernel tests and tuning scripts
DEBUG_SIGNATURES = {
    "ds_debug_conv_set_tile": (C.c_int, [C.c_int, C.c_int]),
    "ds_debug_conv_set_path": (C.c_int, [C.c_int]),
    "ds_debug_conv_set_wide": (C.c_int, [C.c_int]),
    "ds_debug_conv_wino_allow_ablation": (C.c_int, [C.c_int]),
    "ds_debug_conv_wino4_set_nb": (C.c_int, [C.c_int]),
    "ds_debug_conv_bf16_set_max_nb": (C.c_int, [C.c_int]),
    "ds_debug_lstm_seq_set_profile": (C.c_int, [_P]),
    "ds_debug_lstm_seq_set_profile_bwd": (C.c_int, [_P]),
}
TUNING_LIB_PATH = os.path.join(_HERE, "libds_kernels_tuning.so")
_tuning = None


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args


def load_tuning():
    """The tuning build (tests and scripts only; never the product path): the same kernels + the ds_debug_* switches and the
    DS_* environment knobs of the selection rules."""
    global _tuning
    if _tuning is None:
        if not os.path.exists(TUNING_LIB_PATH):
            raise RuntimeError("tumblr_emotions_amd: %s not found -- `make -C tumblr_emotions_amd/csrc` builds it" % TUNING_LIB_PATH)
        lib = C.CDLL(TUNING_LIB_PATH)
        _bind(lib, SIGNATURES)
        _bind(lib, DEBUG_SIGNATURES)
        _tuning = lib
    return _tuning


class tuning_library:
    """with _lib.tuning_library() as lib: every wrapper in ops.py calls into the tuning build inside the block (kernel tests
    that pin a tile, a kernel family or a channel-block count through ds_debug_*); the product library is restored on exit."""

    def __enter__(self):
        global _lib
        self._saved = load()
        _lib = load_tuning()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def load():
    """Load the shared object once; raise loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "tumblr_emotions_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C tumblr_emotions_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    if os.environ.get("DS_LIB"):          # a tuning build replaces the product library: say so once
        import sys
        sys.stderr.write("tumblr_emotions_amd: DS_LIB overrides the kernel library: %s\n" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    _bind(lib, SIGNATURES)
    if hasattr(lib, "ds_debug_conv_set_tile"):      # DS_LIB names a tuning build
        _bind(lib, DEBUG_SIGNATURES)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().ds_last_error().decode()))
