"""TIMING DIAGNOSTIC ONLY (tools/, never imported by the product): FOCAL_ABLATE=<family>[,<family>...] makes the named C-ABI entry
points return FOCAL_OK without launching anything, so that the graph-replayed step can be timed WITHOUT one kernel family -- its
marginal cost under the real two-stream concurrency, which a profiler (it serialises the streams) cannot show.  Outputs of the
skipped kernels stay uninitialised: every result of such a run is garbage, and bench.py marks its JSON line DIAGNOSTIC_ABLATED_INVALID.
Usage: bench.py imports this module only when FOCAL_ABLATE is set (tools/ablate.sh)."""
import os

FAMILIES = {
    "linear_fwd": ["focal_linear_fwd", "focal_linear_resid_ln_fwd"], "linear_bwd_data": ["focal_linear_bwd_data"], "linear_bwd_data_ln": ["focal_linear_bwd_data_ln"],
    "linear_bwd_weight": ["focal_linear_bwd_weight", "focal_linear_bwd_weight_group"], "layernorm_fwd": ["focal_layernorm_fwd"], "layernorm_bwd": ["focal_layernorm_bwd"],
    "mlp_fwd": ["focal_mlp_fwd"], "mlp_bwd": ["focal_mlp_bwd"], "window_attn_fwd": ["focal_window_attn_fwd", "focal_window_attn_qkv_fwd"],
    "window_attn_bwd": ["focal_window_attn_bwd", "focal_window_attn_qkv_bwd"], "conv_fwd": ["focal_conv_fwd"], "conv_bwd_data": ["focal_conv_bwd_data"],
    "conv_bwd_weight": ["focal_conv_bwd_weight"], "bn_stats": ["focal_bn_stats"], "bn_act_fwd": ["focal_bn_act_fwd"],
    "bn_act_bwd": ["focal_bn_act_bwd"], "fft": ["focal_fft_realpack_fwd", "focal_augment_fft_fwd"], "loss_head": ["focal_loss_head"],
    "adamw": ["focal_adamw_multi"], "embed": ["focal_pad_patch_embed_ln_fwd", "focal_pad_patch_embed_ln2_fwd"], "packs": ["focal_pack_multi", "focal_unpack_add_multi"], "gru_seq_fwd": ["focal_gru_seq_fwd"], "gru_seq_bwd": ["focal_gru_seq_bwd"],
    "axpy": ["focal_axpy"], "dropout": ["focal_dropout"], "conv_in": ["focal_conv_in_fwd", "focal_conv_in_bwd_weight"],
}


def install():
    names = [x for x in os.environ.get("FOCAL_ABLATE", "").split(",") if x]
    if not names:
        return []
    from focal_amd import _lib
    lib = _lib.load()
    if "proj64" in names:  # only the 64-channel proj + LayerNorm launches (stage 0): what folding them into the fused MLP kernel could save at most
        names.remove("proj64")
        real = lib.focal_linear_resid_ln_fwd
        setattr(lib, "focal_linear_resid_ln_fwd", lambda d, *a, **k: 0 if d._obj.N == 64 else real(d, *a, **k))
    if "proj_wide" in names:  # the proj launches of stages 1-2 (128 / 256 channels): the bound of folding them into the wide MLP kernel
        names.remove("proj_wide")
        real_ln = lib.focal_linear_resid_ln_fwd
        real_fw = lib.focal_linear_fwd
        setattr(lib, "focal_linear_resid_ln_fwd", lambda d, *a, **k: 0 if d._obj.N == 128 and d._obj.K == 128 else real_ln(d, *a, **k))
        setattr(lib, "focal_linear_fwd", lambda d, *a, **k: 0 if (d._obj.N == 256 and d._obj.K == 256 and d._obj.epilogue == 1) else real_fw(d, *a, **k))
    for n in names:
        for sym in FAMILIES[n]:
            setattr(lib, sym, lambda *a, **k: 0)  # shadows the ctypes function object on this CDLL instance
    return names
