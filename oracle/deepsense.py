"""Functional CPU restatement of the DeepSense backbone (test infrastructure; see oracle/__init__.py).

Dropout layers are identity (parity boundary: stochastic rates forced to 0).  BatchNorm supports
both modes: `train=True` normalises with biased batch statistics and returns the updated running buffers.
"""
import torch
import torch.nn.functional as F


def conv_bn_gelu(P, pre, x, stride, same, train, new_buffers):
    """ConvLayer2D.forward, models/ConvModules.py:98-112 (Dropout2d identity)."""
    w = P[f"{pre}.conv.weight"]
    if same:
        kh, kw = w.shape[2], w.shape[3]
        # torch 'same' padding: total k-1, the extra element (even k) goes on the right/bottom
        lw, lh = (kw - 1) // 2, (kh - 1) // 2
        x = F.pad(x, (lw, kw - 1 - lw, lh, kh - 1 - lh))
    y = F.conv2d(x, w, P[f"{pre}.conv.bias"], stride=stride)
    g, b = P[f"{pre}.batch_norm.weight"], P[f"{pre}.batch_norm.bias"]
    rm, rv = P[f"{pre}.batch_norm.running_mean"], P[f"{pre}.batch_norm.running_var"]
    if train:
        mean = y.mean(dim=(0, 2, 3))
        var_b = y.var(dim=(0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        if new_buffers is not None:
            # momentum 0.1, running_var takes the UNBIASED estimate (ConvModules.py:86)
            new_buffers[f"{pre}.batch_norm.running_mean"] = 0.9 * rm + 0.1 * mean.detach()
            new_buffers[f"{pre}.batch_norm.running_var"] = 0.9 * rv + 0.1 * (var_b.detach() * n / max(n - 1, 1))
    else:
        mean, var_b = rm, rv
    y = (y - mean[None, :, None, None]) * torch.rsqrt(var_b[None, :, None, None] + 1e-5)
    y = y * g[None, :, None, None] + b[None, :, None, None]
    return F.gelu(y)


def conv_block(P, pre, x, in_stride, n_inter, train, new_buffers, taps=None, tap_name=None):
    """ConvBlock.forward, models/ConvModules.py:187-216 -> [B, C_out, intervals]."""
    strided = not (in_stride == 1 or max(in_stride) == 1)
    y = conv_bn_gelu(P, f"{pre}.conv_layer_in", x, tuple(in_stride) if strided else 1, not strided, train, new_buffers)
    if taps is not None:
        taps[f"{tap_name}.conv_in"] = y
    for i in range(n_inter):
        y = y + conv_bn_gelu(P, f"{pre}.conv_layers_inter.{i}", y, 1, True, train, new_buffers)
        if taps is not None:
            taps[f"{tap_name}.inter{i}"] = y
    b, c, i, s = y.shape
    y = y.permute(0, 1, 3, 2).reshape(b, c * s, i)  # channel index = c*S + s (ConvModules.py:207-213)
    return F.conv1d(y, P[f"{pre}.conv_layer_out.weight"], P[f"{pre}.conv_layer_out.bias"])


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.GRU layer, h0 = 0.  x: [B, T, F] -> [B, T, H].  Gate order (r, z, n);
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h  (torch nn.GRU semantics,
    models/RecurrentModule.py:10-12)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = F.linear(x, w_ih, b_ih)  # [B, T, 3H]
    h = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


def recurrent_block(P, pre, x, n_layers):
    """RecurrentBlock.forward, models/RecurrentModule.py:14-31: [B, C, T] -> mean_t bi-GRU output [B, 2H]."""
    y = x.permute(0, 2, 1)
    for layer in range(n_layers):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            outs.append(gru_direction(y, P[f"{pre}.gru.weight_ih_l{layer}{suf}"], P[f"{pre}.gru.weight_hh_l{layer}{suf}"],
                                      P[f"{pre}.gru.bias_ih_l{layer}{suf}"], P[f"{pre}.gru.bias_hh_l{layer}{suf}"], rev))
        y = torch.cat(outs, -1)
    return y.mean(1)


def deepsense_forward(P, cfg, freq_x, proj_head=True, train=False, new_buffers=None, taps=None):
    """DeepSense.forward(freq_x, class_head=False, proj_head=...), models/DeepSense.py:108-166 (1 location)."""
    ds = cfg["DeepSense"]
    assert len(cfg["location_names"]) == 1
    loc = cfg["location_names"][0]
    feats = {}
    for mod in cfg["modality_names"]:
        if isinstance(ds["loc_mod_conv_lens"], dict):
            stride = ds["loc_mod_in_conv_stride"][mod]
        else:
            stride = 1
        y = conv_block(P, f"loc_mod_extractors.{loc}.{mod}", freq_x[loc][mod], stride,
                       ds["loc_mod_conv_inter_layers"], train, new_buffers, taps, f"{loc}.{mod}")
        if taps is not None:
            taps[f"{loc}.{mod}.conv_out"] = y
        feats[mod] = recurrent_block(P, f"recurrent_layers.{mod}", y, ds["recurrent_layers"])
        if taps is not None:
            taps[f"{loc}.{mod}.feat"] = feats[mod]
    if not proj_head:
        return feats
    out = {}
    for mod in cfg["modality_names"]:
        h = F.relu(F.linear(feats[mod], P[f"mod_projectors.{mod}.0.weight"], P[f"mod_projectors.{mod}.0.bias"]))
        out[mod] = F.linear(h, P[f"mod_projectors.{mod}.2.weight"], P[f"mod_projectors.{mod}.2.bias"])
    return out
