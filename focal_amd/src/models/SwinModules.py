"""Swin building blocks of the SW_Transformer backbone -- parameter containers with the reference's names.

Same class names, constructor arguments, parameter / buffer names and registration order as the reference
(models/SwinModules.py), so `state_dict()` interchanges key for key.  The arithmetic does not live here: a whole
(location, modality) encoder is executed by `focal_amd.swin_engine.SwinModEncoder` on the HIP kernels, reading
these parameters through the flat arena.  The buffers (`relative_position_index`, `attn_mask`) are kept for
checkpoint compatibility; the kernels derive both from the window geometry.
"""
import torch
import torch.nn as nn


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop_rate = drop


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
        pts = torch.stack([ys.reshape(-1), xs.reshape(-1)])
        rel = pts[:, :, None] - pts[:, None, :]
        self.register_buffer("relative_position_index", (rel[0] + wh - 1) * (2 * ww - 1) + rel[1] + ww - 1)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


def effective_window(input_resolution, window_size, shift_size):
    """(wh, ww, sh, sw): a window never exceeds the grid, and an axis that fits in one window is not shifted
    (reference SwinModules.py:213-233)."""
    wh, ww = window_size
    sh, sw = shift_size
    if input_resolution[0] <= wh:
        sh, wh = 0, input_resolution[0]
    if input_resolution[1] <= ww:
        sw, ww = 0, input_resolution[1]
    return wh, ww, sh, sw


def region_mask(H, W, wh, ww, sh, sw):
    """[nW, N, N] additive mask of the rolled windows: -100 across wrap-around regions (reference :262-289)."""
    region = torch.zeros(H, W)
    k = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            region[hs, ws] = k
            k += 1
    win = region.view(H // wh, wh, W // ww, ww).permute(0, 2, 1, 3).reshape(-1, wh * ww)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=[5, 5], shift_size=[0, 0], mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, fused_window_process=False):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        wh, ww, sh, sw = effective_window(input_resolution, window_size, shift_size)
        self.window_size, self.shift_size = [wh, ww], [sh, sw]
        self.drop_path_rate = drop_path
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, [wh, ww], num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        # the roll + mask only happen when BOTH axes shift (`min(shift_size) > 0`, reference :262,305)
        self.shifted = min(sh, sw) > 0
        self.register_buffer("attn_mask", region_mask(*input_resolution, wh, ww, sh, sw) if self.shifted else None)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, patch_expanding=None,
                 use_checkpoint=False, fused_window_process=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, list(window_size),
                                 [0, 0] if i % 2 == 0 else [window_size[0] // 2, window_size[1] // 2], mlp_ratio,
                                 qkv_bias, qk_scale, drop, attn_drop,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size=(224, 224), patch_size=[4, 4], in_chans=3, embed_dim=96, norm_layer=None, stride=1):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None
