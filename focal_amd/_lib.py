"""ctypes binding of libfocal_hip.so (the C ABI declared in include/focal_hip.h).

There is no CPU fallback: if the library is missing, `load()` raises, and every op in `focal_amd.ops` goes through
`load()`.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C focal_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FOCAL_HIP_LIB") or os.path.join(_HERE, "libfocal_hip.so")  # (override: A/B runs of two builds on one box)

FOCAL_F32, FOCAL_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU_OUT = 0, 1, 2
EPI_NONE, EPI_RESIDUAL, EPI_RELU, EPI_GELU = 0, 1, 2, 3
BN_EVAL, BN_TRAIN, BN_PARTIAL, BN_FINALIZE = 0, 1, 2, 3
BN_SCRATCH_ZEROED = 16  # OR into the mode: the scratch already holds zeros (no memset launch)
BN_STAT_SLOTS = 16      # focal_conv_fwd_bn: column sums are spread over this many slots of the scratch
ABI_VERSION = 13


class DropDesc(C.Structure):
    _fields_ = [("rng", C.c_void_p), ("stream_elem", C.c_uint32), ("p_elem", C.c_float),
                ("stream_path", C.c_uint32), ("p_path", C.c_float), ("rows_per_sample", C.c_int)]


class FFTDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "I", "n", "n1", "n2")]


class AugDesc(C.Structure):
    _fields_ = [("scale", C.c_float), ("flip", C.c_int), ("use_perm", C.c_int), ("perm", C.c_int * 32), ("phase_cos", C.c_float),
                ("phase_sin", C.c_float)]


class FftProblem(C.Structure):
    _fields_ = [("d", FFTDesc), ("has_aug", C.c_int), ("aug", AugDesc), ("x", C.c_void_p), ("twiddle", C.c_void_p), ("out", C.c_void_p),
                ("plan", C.c_void_p), ("x_warped", C.c_void_p)]


VIEW_MAX_KNOTS, VIEW_MAX_POOL, VIEW_MAX_SLOTS = 16, 8, 8
VIEW_NONE, VIEW_NEGATION, VIEW_SCALING, VIEW_HFLIP, VIEW_PERMUTATION, VIEW_PHASE_SHIFT, VIEW_MAG_WARP, VIEW_TIME_WARP = range(8)


class ViewPlan(C.Structure):
    _fields_ = [("aug", AugDesc), ("kind", C.c_int), ("pool_index", C.c_int), ("warp", C.c_int), ("nknots", C.c_int),
                ("knots", C.c_float * VIEW_MAX_KNOTS)]


class ViewPool(C.Structure):
    _fields_ = [("n_aug", C.c_int), ("kind", C.c_int * VIEW_MAX_POOL), ("prob", C.c_float * VIEW_MAX_POOL), ("scaling_std", C.c_float),
                ("mag_magnitude", C.c_float), ("time_magnitude", C.c_float), ("mag_order", C.c_int), ("time_order", C.c_int),
                ("intervals", C.c_int * VIEW_MAX_SLOTS)]


class WarpProblem(C.Structure):
    _fields_ = [("rows", C.c_int), ("L", C.c_int), ("x", C.c_void_p), ("plan", C.c_void_p), ("tables", C.c_void_p), ("y", C.c_void_p)]


class EmbedDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "cin", "I", "S", "Hp", "Wp", "pw", "C0")] + [("eps", C.c_float)]


class LNDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_int), ("C", C.c_int), ("eps", C.c_float), ("gather", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int)]


class LinearDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("x_dtype", C.c_int),
                ("y_dtype", C.c_int), ("act_in", C.c_int), ("epilogue", C.c_int), ("splits", C.c_int),
                ("out_drop", DropDesc), ("dw_workgroups", C.c_int)]


class DwProblem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("exclusive", C.c_int)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("A", C.c_int), ("B", C.c_int), ("C", C.c_int), ("kind", C.c_int)]


class MlpDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("C", C.c_int), ("hidden", C.c_int), ("drop_hidden", DropDesc),
                ("drop_out", DropDesc), ("ln_eps", C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("heads", C.c_int),
                ("wh", C.c_int), ("ww", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("p_attn", C.c_float),
                ("rng", C.c_void_p), ("stream", C.c_uint32)]


class LossDesc(C.Structure):
    _fields_ = [("n_mod", C.c_int), ("B", C.c_int), ("dim", C.c_int), ("seq", C.c_int), ("temperature", C.c_float),
                ("margin", C.c_float), ("w_shared", C.c_float), ("w_private", C.c_float), ("w_orth", C.c_float),
                ("w_rank", C.c_float), ("no_private", C.c_int)]


class ConvInDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "cin", "I", "S_in", "S_out", "k", "stride", "pad_left", "C")]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "rows", "S", "C_in", "C_out", "k", "dw_workgroups")]


class BNDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_int), ("C", C.c_int), ("rows_per_sample", C.c_int), ("eps", C.c_float),
                ("momentum", C.c_float), ("p_drop", C.c_float), ("rng", C.c_void_p), ("stream", C.c_uint32),
                ("stat_rows", C.c_int), ("groups", C.c_int)]


class GRUDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("H", C.c_int), ("whh_frag", C.c_int)]


class AdamWDesc(C.Structure):
    _fields_ = [("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("l2_decay", C.c_int)]


class TraceRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 384), ("grid", C.c_uint * 3), ("block", C.c_uint * 3), ("stream", C.c_void_p), ("us", C.c_float)]


TRACE_DISPATCH, TRACE_EVENTS = 1, 2
P = C.c_void_p
# name -> (restype, argtypes); every symbol include/focal_hip.h declares
PROTOTYPES = {
    "focal_abi_version": (C.c_int, []),
    "focal_mark": (C.c_int, [P, C.c_int, P]),
    "focal_last_error": (C.c_char_p, []),
    "focal_last_kernel": (C.c_char_p, []),
    "focal_trace_begin": (C.c_int, [C.c_int, C.c_int]),
    "focal_trace_end": (C.c_int, []),
    "focal_trace_count": (C.c_int, []),
    "focal_trace_read": (C.c_int, [C.c_int, C.c_int, C.POINTER(TraceRecord)]),
    "focal_rng_advance": (C.c_int, [P, P]),
    "focal_fft_realpack_fwd": (C.c_int, [C.POINTER(FFTDesc), P, P, P, P]),
    "focal_augment_fft_fwd": (C.c_int, [C.POINTER(FFTDesc), C.POINTER(AugDesc), P, P, P, P]),
    "focal_fft_realpack_multi": (C.c_int, [C.c_int, C.POINTER(FftProblem), P]),
    "focal_warp_fwd": (C.c_int, [C.c_int, C.c_int, P, P, P, P, C.c_int, P, P]),
    "focal_view_draw": (C.c_int, [C.POINTER(ViewPool), C.c_int, C.c_int, P, C.c_uint32, P, P]),
    "focal_view_draw_shared": (C.c_int, [C.POINTER(ViewPool), C.c_int, C.c_int, P, C.c_uint32, P, P]),
    "focal_warp_plan_multi": (C.c_int, [C.c_int, C.POINTER(WarpProblem), P, P]),
    "focal_mixup_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, P, P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    "focal_pad_patch_embed_ln_fwd": (C.c_int, [C.POINTER(EmbedDesc), P, P, P, P, P, P, P]),
    "focal_pad_patch_embed_ln2_fwd": (C.c_int, [C.POINTER(EmbedDesc), P, P, P, P, P, P, P, P, C.c_float, C.c_int, P, P, P]),
    "focal_layernorm_fwd": (C.c_int, [C.POINTER(LNDesc), P, P, P, P, P, P]),
    "focal_layernorm_bwd": (C.c_int, [C.POINTER(LNDesc), P, P, P, P, P, C.c_int, P, P, P, C.POINTER(DropDesc), P]),
    "focal_mask_cast": (C.c_int, [C.c_int, C.c_int, C.c_int, P, C.POINTER(DropDesc), P, P]),
    "focal_linear_fwd": (C.c_int, [C.POINTER(LinearDesc), P, P, P, P, P, P, P]),
    "focal_linear_resid_ln_fwd": (C.c_int, [C.POINTER(LinearDesc), P, P, P, P, P, P, P, C.c_float, P, P, P]),
    "focal_linear_resid_ln_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_linear_bwd_data": (C.c_int, [C.POINTER(LinearDesc), P, P, P, P, P]),
    "focal_linear_bwd_data_ln": (C.c_int, [C.POINTER(LinearDesc), P, P, P, P, P, P, P, P, P, C.POINTER(DropDesc), P]),
    "focal_linear_bwd_data_ln_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_linear_bwd_weight": (C.c_int, [C.POINTER(LinearDesc), P, P, P, P, P]),
    "focal_linear_bwd_weight_workgroups": (C.c_int, [C.POINTER(LinearDesc)]),
    "focal_linear_bwd_weight_kernel": (C.c_int, [C.POINTER(LinearDesc)]),
    "focal_linear_bwd_weight_group": (C.c_int, [C.c_int, C.c_int, C.POINTER(DwProblem), P]),
    "focal_linear_bwd_weight_group_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "focal_linear_bwd_weight_group_workgroups": (C.c_int, [C.c_int, C.c_int, C.POINTER(DwProblem)]),
    "focal_linear_bwd_weight_group_f32": (C.c_int, [C.c_int, C.c_int, C.POINTER(DwProblem), C.c_int, C.c_void_p]),
    "focal_linear_bwd_weight_group_f32_workgroups": (C.c_int, [C.c_int, C.c_int, C.POINTER(DwProblem), C.c_int]),
    "focal_linear_bwd_weight_group_kind": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "focal_mlp_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_mlp_fwd": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "focal_mlp_bwd": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, C.POINTER(DropDesc), P, P, P, P, P]),
    "focal_mlp_bwd_partials_floats": (C.c_long, [C.POINTER(MlpDesc)]),
    "focal_mlp_proj_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_mlp_proj_fwd": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, C.POINTER(DropDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "focal_mlp_wide_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_mlp_wide_fwd": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "focal_mlp_wide_proj_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_mlp_wide_proj_fwd": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, C.POINTER(DropDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "focal_mlp_wide_bwd_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "focal_mlp_wide_bwd_data": (C.c_int, [C.POINTER(MlpDesc), P, P, P, P, P, P, P, P, P, P, P, C.POINTER(DropDesc), P, P, P]),
    "focal_window_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P]),
    "focal_window_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P, P, P]),
    "focal_window_attn_qkv_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "focal_window_attn_qkv_fwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P, P, P]),
    "focal_window_attn_qkv_bwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P, P, P, P, P, P]),
    "focal_fusion_attn_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, C.c_uint32, C.c_float, P]),
    "focal_fusion_attn_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P, P]),
    "focal_cross_entropy": (C.c_int, [C.c_int, C.c_int, P, P, P, P, P]),
    "focal_small_linear_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, P, P, P, P, P]),
    "focal_small_linear_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P]),
    "focal_loss_head_workspace": (C.c_size_t, [C.POINTER(LossDesc)]),
    "focal_loss_head": (C.c_int, [C.POINTER(LossDesc), C.POINTER(P), P, C.POINTER(P), P, C.c_size_t, P]),
    "focal_loss_head_exchange_floats": (C.c_size_t, [C.POINTER(LossDesc), C.c_int]),
    "focal_loss_head_shard_a": (C.c_int, [C.POINTER(LossDesc), C.c_int, C.c_int, C.POINTER(P), P, C.POINTER(P), P, P, C.c_size_t, P]),
    "focal_loss_head_shard_b": (C.c_int, [C.POINTER(LossDesc), C.c_int, C.c_int, C.POINTER(P), P, C.POINTER(P), P, P, C.c_size_t, P]),
    "focal_adamw_multi": (C.c_int, [C.POINTER(AdamWDesc), C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P),
                                    C.POINTER(P), C.POINTER(P), C.POINTER(C.c_long), P, P, P]),
    "focal_adamw_multi_advance": (C.c_int, [C.POINTER(AdamWDesc), C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P),
                                            C.POINTER(P), C.POINTER(P), C.POINTER(C.c_long), P, P, C.c_int, P, P]),
    "focal_cast_bf16": (C.c_int, [P, P, C.c_long, P]),
    "focal_conv_in_fwd": (C.c_int, [C.POINTER(ConvInDesc), P, P, P, P, P]),
    "focal_conv_in_bwd_weight": (C.c_int, [C.POINTER(ConvInDesc), P, P, C.c_int, P, P, P]),
    "focal_permute_pack": (C.c_int, [C.c_int, C.c_int, C.c_int, P, P, C.c_int, P]),
    "focal_permute_unpack_add": (C.c_int, [C.c_int, C.c_int, C.c_int, P, P, P]),
    "focal_pack_multi": (C.c_int, [C.c_int, C.c_int, C.POINTER(PackEntry), P]),
    "focal_unpack_add_multi": (C.c_int, [C.c_int, C.POINTER(PackEntry), P]),
    "focal_conv_pack_bwd": (C.c_int, [C.POINTER(ConvDesc), P, P, P]),
    "focal_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P]),
    "focal_conv_fwd_bn": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, C.POINTER(BNDesc), P, P, P, P, P]),
    "focal_conv_bwd_data": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P]),
    "focal_conv_bwd_weight": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P]),
    "focal_bn_stats": (C.c_int, [C.POINTER(BNDesc), P, P, P, P, P, C.c_int, P]),
    "focal_bn_running_combine": (C.c_int, [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.c_int, C.c_float, P]),
    "focal_bn_act_fwd": (C.c_int, [C.POINTER(BNDesc), P, P, P, P, P, P, P, P]),
    "focal_bn_act_fwd_sums": (C.c_int, [C.POINTER(BNDesc), P, P, P, P, P, P, P, P, P, P, P]),
    "focal_conv_fwd_bn_sums_supported": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(BNDesc), P, P]),
    "focal_bn_act_bwd": (C.c_int, [C.POINTER(BNDesc), P, P, P, P, P, P, P, P, P, C.c_int, P]),
    "focal_gru_gate_fwd": (C.c_int, [C.POINTER(GRUDesc), C.c_int, C.c_int, P, P, P, P, P, P, P]),
    "focal_gru_gate_bwd": (C.c_int, [C.POINTER(GRUDesc), C.c_int, C.c_int, P, C.c_long, C.c_long, C.c_float, P, P, P, P, P, P, P, P]),
    "focal_gru_seq_fwd": (C.c_int, [C.POINTER(GRUDesc), C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), P, P]),
    "focal_gru_seq_bwd": (C.c_int, [C.POINTER(GRUDesc), C.c_int, P, C.c_long, C.c_long, C.c_float, C.POINTER(P), C.POINTER(P), C.POINTER(P),
                                    C.POINTER(P), C.POINTER(P), P]),
    "focal_mean_time": (C.c_int, [C.c_int, C.c_int, C.c_int, P, P, P]),
    "focal_dropout": (C.c_int, [C.c_long, P, P, P, C.c_uint32, C.c_float, P]),
    "focal_axpy": (C.c_int, [C.c_long, C.c_float, P, P, P]),
    "focal_mul": (C.c_int, [C.c_long, P, P, P]),
}

_lib = None


class FocalHipError(RuntimeError):
    pass


def load():
    """Load libfocal_hip.so once; raise (never fall back) if it is absent or its ABI does not match."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FocalHipError(
            f"{LIB_PATH} not found: the FOCAL hot path has no CPU fallback. Build it with "
            "`make -C focal_amd/csrc` (needs hipcc, cross-compiles without a GPU).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.focal_abi_version() != ABI_VERSION:
        raise FocalHipError(f"libfocal_hip ABI {lib.focal_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().focal_last_error()
        raise FocalHipError(f"libfocal_hip error {rc}: {msg.decode() if msg else ''}")
