"""Dataset-config helpers shared by the oracle modules (test infrastructure; see oracle/__init__.py)."""
import math
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_YAML = os.path.join(_HERE, "..", "focal_amd", "src", "data", "MOD.yaml")


def load_config(path=None):
    with open(path or DEFAULT_YAML, "r") as f:
        return yaml.safe_load(f)


def padded_image_size(img_size, window, patch, n_stages):
    """Smallest (H, W) >= img_size such that n_stages-1 halvings of the patch grid still tile by `window`.

    Restates input_utils/padding_utils.py:4-28.
    """
    unit = [window[0] * patch[0] * 2 ** (n_stages - 1), window[1] * patch[1] * 2 ** (n_stages - 1)]
    out = []
    for d in range(2):
        size = max(unit[d], img_size[d])
        out.append(unit[d] * math.ceil(size / unit[d]))
    return out


def swt_geometry(cfg, loc, mod):
    """Per (loc, mod) SW_Transformer geometry: padded image, patch grid, per-stage (H, W, C, depth).

    Restates models/SW_Transformer.py:52-118 (constructor arithmetic only).
    """
    sw = cfg["SW_Transformer"]
    stride = sw["in_stride"][mod]
    spectrum = cfg["loc_mod_spectrum_len"][loc][mod]
    img = (cfg["num_segments"], spectrum // stride)
    patch = sw["patch_size"]["freq"][mod]
    window = sw["window_size"][mod]
    depths = sw["time_freq_block_num"][mod]
    pad_img = padded_image_size(img, window, patch, len(depths))
    grid = [pad_img[0] // patch[0], pad_img[1] // patch[1]]
    c0 = sw["time_freq_out_channels"]
    stages = []
    for i, depth in enumerate(depths):
        stages.append(dict(H=grid[0] // 2 ** i, W=grid[1] // 2 ** i, C=c0 * 2 ** i, depth=depth,
                           downsample=i < len(depths) - 1))
    return dict(stride=stride, img=img, pad_img=pad_img, patch=patch, window=list(window), grid=grid,
                in_chans=cfg["loc_mod_in_freq_channels"][loc][mod] * stride, stages=stages,
                heads=sw["time_freq_head_num"])


def block_window_and_shift(H, W, window, block_idx):
    """Effective (window_h, window_w, shift_h, shift_w, shifted?) of Swin block `block_idx` of a stage.

    Restates models/SwinModules.py:213-233 and the shift_size choice at :470.  Note the quirk: the roll and the
    mask are applied only when BOTH shifts are > 0 (`min(shift) > 0`, :262, :305).
    """
    wh, ww = window
    sh, sw_ = (0, 0) if block_idx % 2 == 0 else (window[0] // 2, window[1] // 2)
    if H <= wh:
        sh, wh = 0, H
    if W <= ww:
        sw_, ww = 0, W
    return wh, ww, sh, sw_, min(sh, sw_) > 0
