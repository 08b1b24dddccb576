"""Padded "image" size for the SW_Transformer patch grid (reference: input_utils/padding_utils.py:4-28)."""
import math


def get_padded_size(img_size, window_size, patch_size, block_nums):
    """Smallest size >= img_size whose patch grid still tiles by `window_size` after block_nums-1 halvings."""
    halvings = 2 ** (block_nums - 1)
    out = []
    for axis in range(2):
        unit = window_size[axis] * patch_size[axis] * halvings
        out.append(unit * math.ceil(max(unit, img_size[axis]) / unit))
    return out
