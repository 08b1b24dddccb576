"""Functional CPU restatement of the SW_Transformer backbone (test infrastructure; see oracle/__init__.py).

State dict in (reference key names), tensors out.  Dropout / DropPath are identity here: the parity boundary is
evaluated with every stochastic rate forced to 0 (train) or in eval mode; their statistics are
tested separately on the HIP path.
"""
import torch
import torch.nn.functional as F

from .config import block_window_and_shift, swt_geometry

# bf16 operand emulation (SURVEY appendix D's method): with `emulate_bf16` the functions below round, in fp32 arithmetic, exactly
# the tensors that the MI355X bf16 path stores or feeds to the matrix cores as bf16 -- linear-layer weights, LayerNorm outputs, qkv,
# the attention probabilities and output, the GELU output -- and nothing else (residual stream, statistics, softmax, biases, patch embedding
# and, since round 6, the operands of the projector stay fp32).  The HIP bf16 path must agree with THIS to a few 1e-3;
# what separates either from the fp32 reference is operand rounding, not arithmetic.
_EMULATE = [False]
# (attribution runs, tests/bf16_error_attribution.py: rounding sites whose name matches one of these regular expressions are left in fp32)
_KEEP_FP32 = []
# round 6: the projector multiplies fp32 operands in the HIP bf16 path (focal_amd/swin_engine.py: tail_fp32); False emulates the round-5
# path, which rounded them
_TAIL_FP32 = [True]


def _r(t, site=""):
    if not _EMULATE[0]:
        return t
    if _TAIL_FP32[0] and site.startswith("mod_projectors."):
        return t
    if _KEEP_FP32:
        import re
        if any(re.search(p, site) for p in _KEEP_FP32):
            return t
    return t.bfloat16().to(t.dtype)


def relative_position_index(wh, ww):
    """[wh*ww, wh*ww] int64 index into the (2wh-1)(2ww-1) bias table.  models/SwinModules.py:101-111."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    pts = torch.stack([ys.reshape(-1), xs.reshape(-1)], 0)  # 2, N
    rel = pts[:, :, None] - pts[:, None, :]  # 2, N, N
    return (rel[0] + wh - 1) * (2 * ww - 1) + (rel[1] + ww - 1)


def shifted_window_mask(H, W, wh, ww, sh, sw):
    """[nW, N, N] additive mask, -100 between tokens of different wrap-around regions.  SwinModules.py:262-289."""
    region = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            region[hs, ws] = cnt
            cnt += 1
    win = region.view(H // wh, wh, W // ww, ww).permute(0, 2, 1, 3).reshape(-1, wh * ww)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def _to_windows(x, wh, ww):
    # [B, H, W, C] -> [B*nW, wh*ww, C]; SwinModules.py:37-52
    B, H, W, C = x.shape
    x = x.view(B, H // wh, wh, W // ww, ww, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, wh * ww, C)


def _from_windows(w, wh, ww, H, W):
    # inverse of _to_windows; SwinModules.py:55-70
    C = w.shape[-1]
    B = w.shape[0] // ((H // wh) * (W // ww))
    x = w.view(B, H // wh, W // ww, wh, ww, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def window_attention(P, pre, xw, heads, wh, ww, mask):
    """WindowAttention.forward, SwinModules.py:121-152 (dropouts are identity)."""
    Bw, N, C = xw.shape
    hd = C // heads
    qkv = _r(F.linear(xw, _r(P[f"{pre}.qkv.weight"], f"{pre}.qkv.weight"), P[f"{pre}.qkv.bias"]), f"{pre}.qkv.out").view(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    table = P[f"{pre}.relative_position_bias_table"]
    bias = table[relative_position_index(wh, ww).reshape(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, heads, N, N) + mask.to(attn.dtype)[None, :, None]).view(-1, heads, N, N)
    attn = _r(attn.softmax(-1), f"{pre}.probs")
    out = _r((attn @ v).transpose(1, 2).reshape(Bw, N, C), f"{pre}.out")
    return F.linear(out, _r(P[f"{pre}.proj.weight"], f"{pre}.proj.weight"), P[f"{pre}.proj.bias"])


def swin_block(P, pre, x, H, W, heads, window, block_idx, taps=None):
    """SwinTransformerBlock.forward, SwinModules.py:294-343."""
    B, L, C = x.shape
    wh, ww, sh, sw, shifted = block_window_and_shift(H, W, window, block_idx)
    y = _r(F.layer_norm(x, (C,), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], 1e-5), f"{pre}.norm1.out").view(B, H, W, C)
    mask = None
    if shifted:
        y = torch.roll(y, shifts=(-sh, -sw), dims=(1, 2))
        mask = shifted_window_mask(H, W, wh, ww, sh, sw)
    aw = window_attention(P, f"{pre}.attn", _to_windows(y, wh, ww), heads, wh, ww, mask)
    y = _from_windows(aw, wh, ww, H, W)
    if shifted:
        y = torch.roll(y, shifts=(sh, sw), dims=(1, 2))
    x = x + y.reshape(B, L, C)
    z = _r(F.layer_norm(x, (C,), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], 1e-5), f"{pre}.norm2.out")
    z = F.linear(z, _r(P[f"{pre}.mlp.fc1.weight"], f"{pre}.mlp.fc1.weight"), P[f"{pre}.mlp.fc1.bias"])
    z = _r(F.gelu(z), f"{pre}.mlp.gelu.out")  # exact erf form, SwinModules.py:19
    z = F.linear(z, _r(P[f"{pre}.mlp.fc2.weight"], f"{pre}.mlp.fc2.weight"), P[f"{pre}.mlp.fc2.bias"])
    return x + z


def patch_merging(P, pre, x, H, W):
    """PatchMerging.forward, SwinModules.py:378-402: 2x2 gather (order x00,x10,x01,x11) -> LN(4C) -> Linear."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = _r(F.layer_norm(x, (4 * C,), P[f"{pre}.norm.weight"], P[f"{pre}.norm.bias"], 1e-5), f"{pre}.norm.out")
    return F.linear(x, _r(P[f"{pre}.reduction.weight"], f"{pre}.reduction.weight"))


def pad_and_embed(P, cfg, x, loc, mod):
    """SW_Transformer.pad_input (:184-208) + PatchEmbed.forward (SwinModules.py:547-558) -> [B, Hp*Wp, C0]."""
    g = swt_geometry(cfg, loc, mod)
    st = g["stride"]
    b, c, i, s = x.shape
    x = x.permute(0, 2, 3, 1).reshape(b, i, s // st, c * st).permute(0, 3, 1, 2)
    x = F.pad(x, (0, g["pad_img"][1] - g["img"][1], 0, g["pad_img"][0] - g["img"][0]))
    pre = f"patch_embed.{loc}.{mod}"
    x = F.conv2d(x, P[f"{pre}.proj.weight"], P[f"{pre}.proj.bias"], stride=tuple(g["patch"]))
    x = x.flatten(2).transpose(1, 2)
    c0 = x.shape[-1]
    return F.layer_norm(x, (c0,), P[f"{pre}.norm.weight"], P[f"{pre}.norm.bias"], 1e-5)


def swt_forward(P, cfg, freq_x, proj_head=True, taps=None, emulate_bf16=False):
    """SW_Transformer.forward(freq_x, class_head=False, proj_head=...), SW_Transformer.py:210-304.

    Returns {mod: [B, emb]} in `modality_names` order.  `taps`, if a dict, receives intermediate activations
    keyed by "<loc>.<mod>.<stage point>" for layer-by-layer parity checks.
    """
    if emulate_bf16 != _EMULATE[0]:
        _EMULATE[0] = emulate_bf16
        try:
            return swt_forward(P, cfg, freq_x, proj_head, taps, emulate_bf16)
        finally:
            _EMULATE[0] = not emulate_bf16
    sw = cfg["SW_Transformer"]
    assert len(cfg["location_names"]) == 1
    loc = cfg["location_names"][0]
    feats = {}
    for mod in cfg["modality_names"]:
        g = swt_geometry(cfg, loc, mod)
        x = pad_and_embed(P, cfg, freq_x[loc][mod], loc, mod)
        if taps is not None:
            taps[f"{loc}.{mod}.embed"] = x
        if sw["APE"]:
            x = x + P[f"absolute_pos_embed.{loc}.{mod}"]
        for si, st in enumerate(g["stages"]):
            for bi in range(st["depth"]):
                x = swin_block(P, f"freq_interval_layers.{loc}.{mod}.{si}.blocks.{bi}", x, st["H"], st["W"],
                               g["heads"], g["window"], bi)
                if taps is not None:
                    taps[f"{loc}.{mod}.s{si}b{bi}"] = x
            if st["downsample"]:
                x = patch_merging(P, f"freq_interval_layers.{loc}.{mod}.{si}.downsample", x, st["H"], st["W"])
                if taps is not None:
                    taps[f"{loc}.{mod}.merge{si}"] = x
        x = F.linear(_r(x.reshape(x.shape[0], -1), f"mod_in_layers.{loc}.{mod}.x"), _r(P[f"mod_in_layers.{loc}.{mod}.weight"], f"mod_in_layers.{loc}.{mod}.weight"),
                     P[f"mod_in_layers.{loc}.{mod}.bias"])
        if taps is not None:
            taps[f"{loc}.{mod}.feat"] = x
        feats[mod] = x
    if not proj_head:
        return feats
    out = {}
    for mod in cfg["modality_names"]:
        h = F.relu(F.linear(_r(feats[mod], f"mod_projectors.{mod}.0.x"), _r(P[f"mod_projectors.{mod}.0.weight"], f"mod_projectors.{mod}.0.weight"),
                            P[f"mod_projectors.{mod}.0.bias"]))
        out[mod] = F.linear(_r(h, f"mod_projectors.{mod}.2.x"), _r(P[f"mod_projectors.{mod}.2.weight"], f"mod_projectors.{mod}.2.weight"), P[f"mod_projectors.{mod}.2.bias"])
    return out
