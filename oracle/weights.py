"""State-dict manifests and name-seeded synthetic weights (test infrastructure; see oracle/__init__.py).

The reference ships no checkpoints and never seeds its RNG, so parity runs fill every floating-point entry of
the state dict from a generator seeded by crc32(key).  The same rule is applied to the reference model by
`tests/golden/gen_golden.py`, to the oracle and to the HIP model, so no weight file has to travel.
"""
import zlib
from collections import OrderedDict

import torch

from .config import block_window_and_shift, swt_geometry


def _fusion_block_spec(prefix, dim, spec):
    # models/FusionModules.py:61-74 (LayerNorm + nn.MultiheadAttention); inert in pretraining.
    spec[f"{prefix}.norm1.weight"] = (dim,)
    spec[f"{prefix}.norm1.bias"] = (dim,)
    spec[f"{prefix}.mha.in_proj_weight"] = (3 * dim, dim)
    spec[f"{prefix}.mha.in_proj_bias"] = (3 * dim,)
    spec[f"{prefix}.mha.out_proj.weight"] = (dim, dim)
    spec[f"{prefix}.mha.out_proj.bias"] = (dim,)


def swt_state_spec(cfg, task="vehicle_classification"):
    """key -> shape of SW_Transformer.state_dict() for a 1-location config (models/SW_Transformer.py:31-182).

    Order follows module registration order in the reference so that it can be compared list-to-list.
    Integer / mask buffers are included (`relative_position_index`, `attn_mask`).
    """
    sw = cfg["SW_Transformer"]
    spec = OrderedDict()
    locs, mods = cfg["location_names"], cfg["modality_names"]
    assert len(locs) == 1, "multi-location fusion is outside the pretraining hot path"
    geo = {(l, m): swt_geometry(cfg, l, m) for l in locs for m in mods}
    # freq_interval_layers
    for l in locs:
        for m in mods:
            g = geo[(l, m)]
            for si, st in enumerate(g["stages"]):
                C = st["C"]
                for bi in range(st["depth"]):
                    p = f"freq_interval_layers.{l}.{m}.{si}.blocks.{bi}"
                    wh, ww, _, _, shifted = block_window_and_shift(st["H"], st["W"], g["window"], bi)
                    n = wh * ww
                    if shifted:
                        spec[f"{p}.attn_mask"] = ((st["H"] // wh) * (st["W"] // ww), n, n)
                    spec[f"{p}.norm1.weight"] = (C,)
                    spec[f"{p}.norm1.bias"] = (C,)
                    spec[f"{p}.attn.relative_position_bias_table"] = ((2 * wh - 1) * (2 * ww - 1), g["heads"])
                    spec[f"{p}.attn.relative_position_index"] = (n, n)
                    spec[f"{p}.attn.qkv.weight"] = (3 * C, C)
                    spec[f"{p}.attn.qkv.bias"] = (3 * C,)
                    spec[f"{p}.attn.proj.weight"] = (C, C)
                    spec[f"{p}.attn.proj.bias"] = (C,)
                    spec[f"{p}.norm2.weight"] = (C,)
                    spec[f"{p}.norm2.bias"] = (C,)
                    hid = 4 * C  # BasicLayer is built with the default mlp_ratio=4.0 (SW_Transformer.py:96-118)
                    spec[f"{p}.mlp.fc1.weight"] = (hid, C)
                    spec[f"{p}.mlp.fc1.bias"] = (hid,)
                    spec[f"{p}.mlp.fc2.weight"] = (C, hid)
                    spec[f"{p}.mlp.fc2.bias"] = (C,)
                if st["downsample"]:
                    p = f"freq_interval_layers.{l}.{m}.{si}.downsample"
                    spec[f"{p}.reduction.weight"] = (2 * C, 4 * C)
                    spec[f"{p}.norm.weight"] = (4 * C,)
                    spec[f"{p}.norm.bias"] = (4 * C,)
    # patch_embed
    for l in locs:
        for m in mods:
            g = geo[(l, m)]
            c0 = g["stages"][0]["C"]
            spec[f"patch_embed.{l}.{m}.proj.weight"] = (c0, g["in_chans"], g["patch"][0], g["patch"][1])
            spec[f"patch_embed.{l}.{m}.proj.bias"] = (c0,)
            spec[f"patch_embed.{l}.{m}.norm.weight"] = (c0,)
            spec[f"patch_embed.{l}.{m}.norm.bias"] = (c0,)
    # absolute_pos_embed
    for l in locs:
        for m in mods:
            g = geo[(l, m)]
            spec[f"absolute_pos_embed.{l}.{m}"] = (1, g["grid"][0] * g["grid"][1], g["stages"][0]["C"])
    # mod_in_layers
    out_c = sw["loc_out_channels"]
    for l in locs:
        for m in mods:
            last = geo[(l, m)]["stages"][-1]
            spec[f"mod_in_layers.{l}.{m}.weight"] = (out_c, last["H"] * last["W"] * last["C"])
            spec[f"mod_in_layers.{l}.{m}.bias"] = (out_c,)
    emb = cfg["FOCAL"]["emb_dim"]
    for m in mods:
        spec[f"mod_projectors.{m}.0.weight"] = (emb, out_c)
        spec[f"mod_projectors.{m}.0.bias"] = (emb,)
        spec[f"mod_projectors.{m}.2.weight"] = (emb, emb)
        spec[f"mod_projectors.{m}.2.bias"] = (emb,)
    _fusion_block_spec("mod_fusion_layers", out_c, spec)
    ncls = cfg[task]["num_classes"]
    if sw["pretrained_head"] == "linear":
        spec["class_layer.0.weight"] = (ncls, out_c)
        spec["class_layer.0.bias"] = (ncls,)
    else:
        spec["class_layer.0.weight"] = (sw["fc_dim"], out_c)
        spec["class_layer.0.bias"] = (sw["fc_dim"],)
        spec["class_layer.2.weight"] = (ncls, sw["fc_dim"])
        spec["class_layer.2.bias"] = (ncls,)
    return spec


def _conv_block_spec(prefix, cin, cout, spectrum, conv_lens, n_inter, in_stride, spec):
    # models/ConvModules.py:115-185
    half = cout // 2

    def layer(p, ci, co, k):
        spec[f"{p}.conv.weight"] = (co, ci, k[0], k[1])
        spec[f"{p}.conv.bias"] = (co,)
        spec[f"{p}.batch_norm.weight"] = (co,)
        spec[f"{p}.batch_norm.bias"] = (co,)
        spec[f"{p}.batch_norm.running_mean"] = (co,)
        spec[f"{p}.batch_norm.running_var"] = (co,)
        spec[f"{p}.batch_norm.num_batches_tracked"] = ()

    layer(f"{prefix}.conv_layer_in", cin, half, conv_lens[0])
    for i in range(n_inter):
        layer(f"{prefix}.conv_layers_inter.{i}", half, half, conv_lens[1])
    s_out = spectrum if in_stride == 1 else spectrum // in_stride[1]
    spec[f"{prefix}.conv_layer_out.weight"] = (cout, half * s_out, 1)
    spec[f"{prefix}.conv_layer_out.bias"] = (cout,)


def deepsense_state_spec(cfg, task="vehicle_classification"):
    """key -> shape of DeepSense.state_dict() (models/DeepSense.py:33-106)."""
    ds = cfg["DeepSense"]
    spec = OrderedDict()
    locs, mods = cfg["location_names"], cfg["modality_names"]
    for l in locs:
        for m in mods:
            if isinstance(ds["loc_mod_conv_lens"], dict):
                lens, stride = ds["loc_mod_conv_lens"][m], ds["loc_mod_in_conv_stride"][m]
            else:
                lens, stride = ds["loc_mod_conv_lens"], 1
            stride = 1 if (stride == 1 or max(stride) == 1) else stride
            _conv_block_spec(f"loc_mod_extractors.{l}.{m}", cfg["loc_mod_in_freq_channels"][l][m],
                             ds["loc_mod_out_channels"], cfg["loc_mod_spectrum_len"][l][m], lens,
                             ds["loc_mod_conv_inter_layers"], stride, spec)
    for m in mods:
        _conv_block_spec(f"mod_extractors.{m}", 1, ds["loc_out_channels"], ds["loc_mod_out_channels"],
                         ds["loc_conv_lens"], ds["loc_conv_inter_layers"], 1, spec)
    H = ds["recurrent_dim"]
    for m in mods:
        for layer in range(ds["recurrent_layers"]):
            fin = ds["loc_out_channels"] if layer == 0 else 2 * H
            for suf in ("", "_reverse"):
                spec[f"recurrent_layers.{m}.gru.weight_ih_l{layer}{suf}"] = (3 * H, fin)
                spec[f"recurrent_layers.{m}.gru.weight_hh_l{layer}{suf}"] = (3 * H, H)
                spec[f"recurrent_layers.{m}.gru.bias_ih_l{layer}{suf}"] = (3 * H,)
                spec[f"recurrent_layers.{m}.gru.bias_hh_l{layer}{suf}"] = (3 * H,)
    emb = cfg["FOCAL"]["emb_dim"]
    for m in mods:
        spec[f"mod_projectors.{m}.0.weight"] = (emb, 2 * H)
        spec[f"mod_projectors.{m}.0.bias"] = (emb,)
        spec[f"mod_projectors.{m}.2.weight"] = (emb, emb)
        spec[f"mod_projectors.{m}.2.bias"] = (emb,)
    sample_dim = 2 * H * len(mods)
    ncls = cfg[task]["num_classes"]
    if ds["pretrained_head"] == "linear":
        spec["class_layer.0.weight"] = (ncls, sample_dim)
        spec["class_layer.0.bias"] = (ncls,)
    else:
        spec["class_layer.0.weight"] = (ds["fc_dim"], sample_dim)
        spec["class_layer.0.bias"] = (ds["fc_dim"],)
        spec["class_layer.2.weight"] = (ncls, ds["fc_dim"])
        spec["class_layer.2.bias"] = (ncls,)
    return spec


NON_FLOAT_SUFFIXES = ("relative_position_index", "num_batches_tracked", "attn_mask")


def seeded_values(key, shape, scale_override=None):
    """Deterministic fp32 values for state-dict entry `key` (generator seeded with crc32 of the name)."""
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    r = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    leaf = key.rsplit(".", 1)[-1]
    if scale_override is not None:
        return r * scale_override
    if leaf == "running_var":
        return 1.0 + 0.25 * r.abs()
    if leaf == "running_mean":
        return 0.1 * r
    if leaf == "relative_position_bias_table":
        return 0.5 * r
    if leaf.startswith("bias") or leaf == "in_proj_bias":
        return 0.05 * r
    if "norm" in key.rsplit(".", 2)[-2] and leaf == "weight":
        return 1.0 + 0.1 * r  # LayerNorm / BatchNorm scale
    if leaf.startswith("weight_hh") or leaf.startswith("weight_ih"):
        return r / (shape[1] ** 0.5)
    if key.startswith("absolute_pos_embed"):
        return 0.02 * r
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return r / (fan_in ** 0.5)
    return 0.05 * r


def fill_state_dict_(state):
    """Overwrite every floating-point entry of `state` (name -> tensor) in place with seeded_values."""
    with torch.no_grad():
        for k, v in state.items():
            if k.endswith(NON_FLOAT_SUFFIXES) or not v.is_floating_point():
                continue
            v.copy_(seeded_values(k, v.shape).to(v.dtype))
    return state


def synthetic_freq_input(cfg, batch, seed, dtype=torch.float32, scale_like_fft=True):
    """Seeded frequency-domain batch {loc: {mod: [B, c, i, s]}} (the parity boundary input)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for l in cfg["location_names"]:
        out[l] = {}
        for m in cfg["modality_names"]:
            c = cfg["loc_mod_in_freq_channels"][l][m]
            s = cfg["loc_mod_spectrum_len"][l][m]
            x = torch.randn(batch, c, cfg["num_segments"], s, generator=g, dtype=torch.float32)
            if scale_like_fft:
                x = x * (s ** 0.5) * 0.7071  # magnitude of an FFT bin of N(0,1) samples
            out[l][m] = x.to(dtype)
    return out


def synthetic_time_input(cfg, batch, seed):
    """Seeded time-domain batch {loc: {mod: [B, c_time, i, s]}}, N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for l in cfg["location_names"]:
        out[l] = {}
        for m in cfg["modality_names"]:
            c = cfg["loc_mod_in_time_channels"][l][m]
            s = cfg["loc_mod_spectrum_len"][l][m]
            out[l][m] = torch.randn(batch, c, cfg["num_segments"], s, generator=g, dtype=torch.float32)
    return out
