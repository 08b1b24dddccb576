"""Tensor-level wrappers over the libfocal_hip C ABI.

PyTorch is plumbing here: it owns device memory (torch.empty), the current HIP stream and autograd bookkeeping; all
arithmetic happens in the HIP kernels.  Every function enqueues on `torch.cuda.current_stream()` and returns
immediately; none of them has a CPU implementation.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU_OUT, EPI_GELU, EPI_NONE, EPI_RELU, EPI_RESIDUAL, FOCAL_BF16, FOCAL_F32,
                   AdamWDesc, AttnDesc, BNDesc, ConvDesc, ConvInDesc, DropDesc, EmbedDesc, FFTDesc, GRUDesc, LinearDesc,
                   LNDesc, LossDesc, MlpDesc, check)

_TORCH2CODE = {torch.float32: FOCAL_F32, torch.bfloat16: FOCAL_BF16}
_CODE2TORCH = {v: k for k, v in _TORCH2CODE.items()}


def code(dtype):
    return _TORCH2CODE[dtype]


def torch_dtype(c):
    return _CODE2TORCH[c]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------ per-step zero pool
# Many kernels accumulate with atomics into small buffers that must start at zero (BatchNorm channel sums, split weight
# gradients).  One memset per buffer is one graph node each (~100 per DeepSense step); instead they are slices of ONE fp32
# pool that `zero_pool_reset` (called by the optimizer's zero_grad, after the side streams are joined) zeroes with a single
# launch.  A slice is handed out once between two resets, so correctness never depends on the reset being called: without it
# the pool simply runs out and callers fall back to their own zero-fill.
_ZERO_POOL = {}
_ZERO_POOL_FLOATS = 4 << 20


def zero_pool_reset(device):
    st = _ZERO_POOL.get(torch.device(device))
    if st is None:
        return
    if st["ptr"] > 0:
        st["buf"][:st["ptr"]].zero_()
    st["ptr"] = 0


def pool_zeros(n, device):
    """n zero fp32 values from the pool, or None when the pool cannot serve them (caller then zero-fills itself)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return None
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _ZERO_POOL.get(dev)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None  # never create the pool inside a graph capture
        st = _ZERO_POOL[dev] = {"buf": torch.zeros(_ZERO_POOL_FLOATS, dtype=torch.float32, device=dev), "ptr": 0}
        torch.cuda.current_stream(dev).synchronize()  # created on whichever stream asked first; every stream must see the zeros
    need = (n + 63) // 64 * 64
    if st["ptr"] + need > _ZERO_POOL_FLOATS:
        return None
    out = st["buf"][st["ptr"]:st["ptr"] + n]
    st["ptr"] += need
    return out


def zeros(shape, device):
    """fp32 zeros of `shape`: a pool slice when possible, else torch.zeros."""
    n = 1
    for v in shape:
        n *= int(v)
    t = pool_zeros(n, device)
    return t.view(*shape) if t is not None else torch.zeros(*shape, dtype=torch.float32, device=device)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.FocalHipError("libfocal_hip operates on device tensors only (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise _lib.FocalHipError("libfocal_hip needs contiguous tensors")


def drop_desc(rng=None, stream_elem=0, p_elem=0.0, stream_path=0, p_path=0.0, rows_per_sample=1):
    return DropDesc(_p(rng), stream_elem, p_elem, stream_path, p_path, rows_per_sample)


NO_DROP = drop_desc()


# ------------------------------------------------------------------------------------------------ RNG / optimizer
def new_rng_state(seed, device):
    return torch.tensor([seed & 0xFFFFFFFF, 0, 0, 0], dtype=torch.int64, device=device).to(torch.int32)


def new_step_state(device, count=0):
    """The optimizer's device-side step state for adamw_multi(advance=True): {seed, step count, tickets ...} (FOCAL_STEP_STATE_WORDS)."""
    st = torch.zeros(40, dtype=torch.int32, device=device)
    if count:
        st[1] = int(count)
    return st


def mark(slots, index):
    """Diagnostic: slots[index] (int64 device tensor) = device wall clock, 100 MHz ticks, when the launch runs on the current stream."""
    check(_lib.load().focal_mark(_p(slots), int(index), _stream()))


def rng_advance(state):
    check(_lib.load().focal_rng_advance(_p(state), _stream()))


def cast_bf16(src, dst):
    _need_cuda(src, dst)
    check(_lib.load().focal_cast_bf16(_p(src), _p(dst), src.numel(), _stream()))


def adamw_multi(segments, lr_dev, rng_state, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, l2_decay=False, advance=False, seed_state=None):
    """segments: list of (p, g, m, v, shadow_or_None) flat tensors (lengths multiples of 4).  rng_state[1] is the step count the update
    uses; advance=True: the count used is rng_state[1] + 1 and the kernel itself then advances rng_state and (if given) seed_state
    (focal_adamw_multi_advance: what two focal_rng_advance launches in front of the call would have done)."""
    n = len(segments)
    arr = lambda i: (C.c_void_p * n)(*[_p(s[i]) for s in segments])
    has_shadow = any(s[4] is not None for s in segments)
    lens = (C.c_long * n)(*[s[0].numel() for s in segments])
    d = AdamWDesc(beta1, beta2, eps, weight_decay, int(l2_decay))
    if advance:
        check(_lib.load().focal_adamw_multi_advance(C.byref(d), n, arr(0), arr(1), arr(2), arr(3), arr(4) if has_shadow else None,
                                                    lens, _p(lr_dev), _p(rng_state), rng_state.numel(), _p(seed_state), _stream()))
        return
    check(_lib.load().focal_adamw_multi(C.byref(d), n, arr(0), arr(1), arr(2), arr(3), arr(4) if has_shadow else None,
                                        lens, _p(lr_dev), _p(rng_state), _stream()))


# ------------------------------------------------------------------------------------------------ row 3
_TWIDDLES = {}


def _factor(n):
    """n = n1 * n2 with n1 as close to sqrt(n) as possible (n2 = 1 -> direct DFT for short rows)."""
    if n <= 64:
        return n, 1
    best = 1
    for a in range(1, int(math.isqrt(n)) + 1):
        if n % a == 0:
            best = a
    return n // best, best


def _fft_problem(x, scale=1.0, flip=False, perm=None, phase=0.0, out=None, plan=None, x_warped=None):
    """(descriptor, augmentation descriptor | None, x, twiddle table, out) of one transform.  plan (a row of ops.view_draw's tensor) /
    x_warped are bound by fft_realpack_multi."""
    _need_cuda(x)
    B, Cc, I, n = x.shape
    key = (n, x.device)
    if key not in _TWIDDLES:
        k = torch.arange(n, dtype=torch.float64)
        ang = 2.0 * math.pi * k / n
        _TWIDDLES[key] = torch.stack([torch.cos(ang), -torch.sin(ang)], 1).to(torch.float32).contiguous().to(x.device)
    n1, n2 = _factor(n)
    if out is None:
        out = torch.empty(B, 2 * Cc, I, n, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, 2 * Cc, I, n) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("fft_realpack: `out` must be a contiguous fp32 [B, 2C, I, n] tensor on x's device")
    d = FFTDesc(B, Cc, I, n, n1, n2)
    if scale == 1.0 and not flip and perm is None and phase == 0.0:
        return d, None, x, _TWIDDLES[key], out
    a = _lib.AugDesc()
    a.scale, a.flip, a.use_perm = float(scale), int(bool(flip)), int(perm is not None)
    a.phase_cos, a.phase_sin = math.cos(phase), math.sin(phase)
    if perm is not None:
        perm = [int(v) for v in perm]
        if sorted(perm) != list(range(I)):
            raise ValueError(f"perm must be a permutation of range({I})")
        for i, v in enumerate(perm):
            a.perm[i] = v
    return d, a, x, _TWIDDLES[key], out


def fft_realpack(x, scale=1.0, flip=False, perm=None, phase=0.0, out=None):
    """[B, C, I, n] real fp32 -> [B, 2C, I, n] (Re/Im channel pairs of the full two-sided spectrum).

    Optional view augmentation folded into the transform (focal_augment_fft_fwd): x * scale, horizontal flip (intervals and
    samples reversed), interval order `perm` (sequence of I ints), and a rotation of every bin by `phase` radians.
    `out`: write into this contiguous [B, 2C, I, n] fp32 tensor (e.g. one half of a two-view batch) instead of allocating."""
    d, a, x, tw, out = _fft_problem(x, scale, flip, perm, phase, out)
    if a is None:
        check(_lib.load().focal_fft_realpack_fwd(C.byref(d), _p(x), _p(tw), _p(out), _stream()))
    else:
        check(_lib.load().focal_augment_fft_fwd(C.byref(d), C.byref(a), _p(x), _p(tw), _p(out), _stream()))
    return out


def fft_realpack_multi(items):
    """items: [dict(x=..., scale=, flip=, perm=, phase=, out=)] -- the keyword arguments of fft_realpack, one dict per transform (the
    modalities of a view, or of both views) -> list of outputs.  One call (focal_fft_realpack_multi): the short-row transforms (the
    20-sample sensor modalities) share one launch, the others are launched as fft_realpack would."""
    probs = [_fft_problem(**it) for it in items]
    arr = (_lib.FftProblem * len(probs))()
    for i, (d, a, x, tw, out) in enumerate(probs):
        arr[i].d = d
        arr[i].has_aug = int(a is not None)
        if a is not None:
            arr[i].aug = a
        arr[i].x, arr[i].twiddle, arr[i].out = _p(x), _p(tw), _p(out)
        plan = items[i].get("plan")
        if plan is not None:  # the augmentation is read from the device record when the kernel runs (view draws inside the captured step)
            _need_cuda(plan, items[i]["x_warped"])
            arr[i].plan, arr[i].x_warped = _p(plan), _p(items[i]["x_warped"])
    check(_lib.load().focal_fft_realpack_multi(len(probs), arr, _stream()))
    return [p[4] for p in probs]


# ---- view draws on the device (focal_view_draw / focal_warp_plan_multi: include/focal_hip.h)
VIEW_KINDS = {"no": _lib.VIEW_NONE, "negation": _lib.VIEW_NEGATION, "scaling": _lib.VIEW_SCALING, "horizontal_flip": _lib.VIEW_HFLIP,
              "permutation": _lib.VIEW_PERMUTATION, "phase_shift": _lib.VIEW_PHASE_SHIFT, "mag_warp": _lib.VIEW_MAG_WARP,
              "time_warp": _lib.VIEW_TIME_WARP}
VIEW_PLAN_BYTES = C.sizeof(_lib.ViewPlan)


def view_pool(entries, intervals, scaling_std=0.2, mag_warp=(0.05, 4), time_warp=(0.2, 6)):
    """entries: [(augmenter name, coin probability)] -- the pool ONE entry of which is drawn per view; intervals: per slot ((location,
    modality) pair) the interval count a permutation shuffles; the two warps' (magnitude, spline ord)."""
    if not 1 <= len(entries) <= _lib.VIEW_MAX_POOL or not 1 <= len(intervals) <= _lib.VIEW_MAX_SLOTS:
        raise ValueError("view_pool: 1 .. 8 pool entries, 1 .. 8 slots")
    pool = _lib.ViewPool()
    pool.n_aug = len(entries)
    for i, (name, prob) in enumerate(entries):
        pool.kind[i], pool.prob[i] = VIEW_KINDS[name], float(prob)
    pool.scaling_std = float(scaling_std)
    pool.mag_magnitude, pool.mag_order = float(mag_warp[0]), int(mag_warp[1])
    pool.time_magnitude, pool.time_order = float(time_warp[0]), int(time_warp[1])
    for i, n in enumerate(intervals):
        pool.intervals[i] = int(n)
    return pool


def new_view_plans(n_views, n_slots, device):
    """uint8 [n_views * n_slots, sizeof(focal_view_plan)]: row v * n_slots + s is the plan of (view v, slot s)."""
    return torch.zeros(n_views * n_slots, VIEW_PLAN_BYTES, dtype=torch.uint8, device=device)


def view_draw(pool, n_views, n_slots, seed_state, stream_id, plans):
    """The draws of n_views views over n_slots slots into `plans` (new_view_plans), keyed by seed_state[0] (the device seed word the
    optimizer advances every step) and stream_id."""
    _need_cuda(plans, seed_state)
    assert plans.shape == (n_views * n_slots, VIEW_PLAN_BYTES) and plans.dtype == torch.uint8 and plans.is_contiguous()
    check(_lib.load().focal_view_draw(C.byref(pool), n_views, n_slots, _p(seed_state), int(stream_id) & 0xFFFFFFFF, _p(plans), _stream()))
    return plans


def view_draw_shared(pool, n_views, n_slots, view_state, stream_id, plans):
    """view_draw from the draw's own device state (runtime.view_state: one seed broadcast to every data-parallel rank), which the kernel
    advances itself: every rank draws the same plans, replay after replay."""
    _need_cuda(plans, view_state)
    assert plans.shape == (n_views * n_slots, VIEW_PLAN_BYTES) and plans.dtype == torch.uint8 and plans.is_contiguous()
    assert view_state.dtype == torch.int32 and view_state.numel() >= 4
    check(_lib.load().focal_view_draw_shared(C.byref(pool), n_views, n_slots, _p(view_state), int(stream_id) & 0xFFFFFFFF, _p(plans), _stream()))
    return plans


def read_view_plans(plans):
    """Host copies of the plan records (tests, diagnostics): a list of _lib.ViewPlan."""
    raw = plans.cpu().numpy().tobytes()
    return [_lib.ViewPlan.from_buffer_copy(raw[i * VIEW_PLAN_BYTES:(i + 1) * VIEW_PLAN_BYTES]) for i in range(plans.shape[0])]


def write_view_plan(plans, index, scale=1.0, flip=False, perm=None, phase=0.0, warp=0, knots=None, kind=0):
    """Force one plan record (tests: the reference fixtures' forced draws through the device-plan path)."""
    pl = _lib.ViewPlan()
    pl.aug.scale, pl.aug.flip, pl.aug.use_perm = float(scale), int(bool(flip)), int(perm is not None)
    pl.aug.phase_cos, pl.aug.phase_sin = math.cos(phase), math.sin(phase)
    for i in range(32):
        pl.aug.perm[i] = int(perm[i]) if perm is not None and i < len(perm) else i
    pl.kind, pl.warp = int(kind), int(warp)
    if int(warp) != 0 and (knots is None or not 4 <= len(knots) <= _lib.VIEW_MAX_KNOTS):
        raise ValueError(f"write_view_plan: a warp needs 4 .. {_lib.VIEW_MAX_KNOTS} knots")
    if knots is not None:
        pl.nknots = len(knots)
        for i, v in enumerate(knots):
            pl.knots[i] = float(v)
    buf = torch.frombuffer(bytearray(bytes(pl)), dtype=torch.uint8)
    plans[index].copy_(buf)


_END_COEF = {}


def warp_end_coefficients(device):
    """Device fp32 [47][4][48]: the end-window basis splines of the time warp (focal_amd.warp.end_window_coefficients), once per device."""
    key = torch.device(device)
    if key not in _END_COEF:
        from . import warp
        _END_COEF[key] = torch.from_numpy(warp.end_window_coefficients()).to(device)
    return _END_COEF[key]


def warp_plan_multi(problems):
    """problems: [dict(x=[B, C, I, S] fp32, plan=row of the plan tensor, tables=fp32 [2 * I * S] workspace, y=like x)]: the warps the
    plans ask for (focal_warp_plan_multi: two launches whatever they ask)."""
    if len(problems) > 8:  # (the launch table holds 8 problems: a dataset with 5 .. 8 (location, modality) slots asks for 10 .. 16)
        for k in range(0, len(problems), 8):
            warp_plan_multi(problems[k:k + 8])
        return
    arr = (_lib.WarpProblem * len(problems))()
    for i, q in enumerate(problems):
        x, y = q["x"], q["y"]
        _need_cuda(x, y, q["plan"], q["tables"])
        B, Cc, I, S = x.shape
        assert y.shape == x.shape and x.is_contiguous() and y.is_contiguous() and q["tables"].numel() >= 2 * I * S
        arr[i].rows, arr[i].L = B * Cc, I * S
        arr[i].x, arr[i].plan, arr[i].tables, arr[i].y = _p(x), _p(q["plan"]), _p(q["tables"]), _p(y)
    check(_lib.load().focal_warp_plan_multi(len(problems), arr, _p(warp_end_coefficients(problems[0]["x"].device)), _stream()))


def mag_warp(x, mult):
    """[B, C, I, S] fp32 x mult [I*S] (device fp32): the magnitude warp of focal_warp_fwd."""
    _need_cuda(x, mult)
    B, Cc, I, S = x.shape
    y = torch.empty_like(x)
    check(_lib.load().focal_warp_fwd(B * Cc, I * S, _p(x), _p(mult), None, None, 0, _p(y), _stream()))
    return y


def time_warp(x, k0, w):
    """[B, C, I, S] fp32 resampled along the flattened (I, S) axis with the tables of focal_amd.warp.time_warp_tables (device)."""
    _need_cuda(x, k0, w)
    B, Cc, I, S = x.shape
    y = torch.empty_like(x)
    check(_lib.load().focal_warp_fwd(B * Cc, I * S, _p(x), None, _p(k0), _p(w), w.shape[1], _p(y), _stream()))
    return y


def mixup(x, perm, lam=1.0, box=None):
    """focal_mixup_fwd: x fp32 [B, C, I, S], perm int32 [B] (device).  box = (yl, yh, xl, xh) -> CutMix paste, else lam-blend."""
    _need_cuda(x, perm)
    B, Cc, I, S = x.shape
    y = torch.empty_like(x)
    yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
    check(_lib.load().focal_mixup_fwd(B, Cc, I, S, _p(x), _p(perm), float(lam), int(box is not None), int(yl), int(yh), int(xl), int(xh),
                                      _p(y), _stream()))
    return y


# ------------------------------------------------------------------------------------------------ row 8
def pad_patch_embed_ln(x, w, b, gamma, beta, Hp, Wp, pw, eps=1e-5, next_ln=None):
    """next_ln = (gamma2, beta2, out_dtype): also return (LayerNorm(tokens), stats) of the LayerNorm that follows (C0 == 64)."""
    _need_cuda(x, w, b, gamma, beta)
    B, cin, I, S = x.shape
    C0 = w.shape[0]
    out = torch.empty(B * Hp * Wp, C0, dtype=torch.float32, device=x.device)
    d = EmbedDesc(B, cin, I, S, Hp, Wp, pw, C0, eps)
    if next_ln is None:
        check(_lib.load().focal_pad_patch_embed_ln_fwd(C.byref(d), _p(x), _p(w), _p(b), _p(gamma), _p(beta), _p(out), _stream()))
        return out
    g2, b2, dt = next_ln
    y_ln = torch.empty(B * Hp * Wp, C0, dtype=dt, device=x.device)
    stats = torch.empty(B * Hp * Wp, 2, dtype=torch.float32, device=x.device)
    check(_lib.load().focal_pad_patch_embed_ln2_fwd(C.byref(d), _p(x), _p(w), _p(b), _p(gamma), _p(beta), _p(out), _p(g2), _p(b2),
                                                    1e-5, code(dt), _p(y_ln), _p(stats), _stream()))
    return out, y_ln, stats


# ------------------------------------------------------------------------------------------------ LayerNorm
def ln_desc(dtype_code, rows, Cc, eps=1e-5, gather=None):
    if gather is None:
        return LNDesc(dtype_code, rows, Cc, eps, 0, 0, 0, 0, 0)
    B, H, W, Cin = gather
    return LNDesc(dtype_code, rows, Cc, eps, 1, B, H, W, Cin)


def layernorm_fwd(x, gamma, beta, out_dtype, gather=None, desc=None):
    """x fp32 [rows, C] (or [B, H, W, Cin] with gather=(B, H, W, Cin)) -> (y [rows, C] out_dtype, stats [rows, 2])."""
    _need_cuda(x, gamma, beta)
    if gather is None:
        rows, Cc = x.shape
    else:
        B, H, W, Cin = gather
        rows, Cc = B * (H // 2) * (W // 2), 4 * Cin
    d = desc or ln_desc(code(out_dtype), rows, Cc, gather=gather)
    y = torch.empty(rows, Cc, dtype=out_dtype, device=x.device)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    check(_lib.load().focal_layernorm_fwd(C.byref(d), _p(x), _p(gamma), _p(beta), _p(y), _p(stats), _stream()))
    return y, stats


def layernorm_bwd(dy, x, stats, gamma, dx, accumulate, dgamma, dbeta, gather=None, desc=None, dx_masked=None, mask=None):
    """dx_masked (dy.dtype, same shape as dx) <- dx * mask for the residual branch that consumes dx next (mask=None: a
    plain dtype copy)."""
    _need_cuda(dy, x, stats, gamma, dx, dgamma, dbeta)
    rows, Cc = dy.shape
    d = desc or ln_desc(code(dy.dtype), rows, Cc, gather=gather)
    if dx_masked is not None:
        assert dx_masked.dtype == dy.dtype and dx_masked.numel() == dx.numel()
        mask = mask or NO_DROP
    check(_lib.load().focal_layernorm_bwd(C.byref(d), _p(dy), _p(x), _p(stats), _p(gamma), _p(dx), int(accumulate),
                                          _p(dgamma), _p(dbeta), _p(dx_masked), C.byref(mask) if mask is not None else None,
                                          _stream()))


def mask_cast(g, mask, dtype):
    """dtype(g * mask) for an fp32 [rows, C] gradient (see focal_mask_cast)."""
    _need_cuda(g)
    rows, Cc = g.shape
    out = torch.empty(rows, Cc, dtype=dtype, device=g.device)
    check(_lib.load().focal_mask_cast(code(dtype), rows, Cc, _p(g), C.byref(mask or NO_DROP), _p(out), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ Linear family
def linear_desc(dtype_code, M, N, K, x_dtype, y_dtype, act_in=ACT_NONE, epilogue=EPI_NONE, splits=1, out_drop=None, dw_workgroups=0):
    return LinearDesc(dtype_code, M, N, K, x_dtype, y_dtype, act_in, epilogue, splits, out_drop or NO_DROP, dw_workgroups)


def linear_fwd(d, x, w, bias, resid, y, act_grad=None):
    check(_lib.load().focal_linear_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(resid), _p(y), _p(act_grad), _stream()))


def linear_resid_ln_fwd(d, x, w, bias, resid, y, gamma, beta, out_dtype, eps=1e-5):
    """y = resid + drop(x w^T + bias) and, from the same kernel, (LayerNorm(y), stats) of the Swin LayerNorm that follows
    (N == 64; N == 128 / 256 in bf16: ops.resid_ln_supported).  Returns (y_ln [M, N] out_dtype, stats [M, 2])."""
    y_ln = torch.empty(d.M, d.N, dtype=out_dtype, device=y.device)
    stats = torch.empty(d.M, 2, dtype=torch.float32, device=y.device)
    check(_lib.load().focal_linear_resid_ln_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(resid), _p(y), _p(gamma), _p(beta), eps,
                                                _p(y_ln), _p(stats), _stream()))
    return y_ln, stats


def resid_ln_supported(dtype_code, N, K):
    return bool(_lib.load().focal_linear_resid_ln_supported(dtype_code, N, K))


def linear_bwd_data(d, dy, w, x, dx):
    check(_lib.load().focal_linear_bwd_data(C.byref(d), _p(dy), _p(w), _p(x), _p(dx), _stream()))


def bwd_data_ln_supported(dtype_code, N, K):
    return bool(_lib.load().focal_linear_bwd_data_ln_supported(dtype_code, N, K))


def linear_bwd_data_ln(d, dy, w, ln_x, ln_stats, ln_gamma, g, dgamma, dbeta, g_masked=None, mask=None):
    """dX of the linear layer `d` and the backward of the LayerNorm that produced its input, in one kernel: g += dLN(dy . w); dgamma / dbeta
    accumulate; g_masked (dy.dtype) <- g * mask for the residual branch that consumes g next (mask=None: a plain cast)."""
    _need_cuda(dy, w, ln_x, ln_stats, ln_gamma, g, dgamma, dbeta, g_masked)
    if g_masked is not None:
        mask = mask or NO_DROP
    check(_lib.load().focal_linear_bwd_data_ln(C.byref(d), _p(dy), _p(w), _p(ln_x), _p(ln_stats), _p(ln_gamma), _p(g), _p(dgamma), _p(dbeta),
                                               _p(g_masked), C.byref(mask) if mask is not None else None, _stream()))


def linear_bwd_weight(d, dy, x, dw, dbias):
    check(_lib.load().focal_linear_bwd_weight(C.byref(d), _p(dy), _p(x), _p(dw), _p(dbias), _stream()))


def dw_group_supported(dtype_code, M, N, K):
    return bool(_lib.load().focal_linear_bwd_weight_group_supported(dtype_code, M, N, K))


def dw_group_kind(dtype_code, M, N, K):
    """2: a shape for the 128 x 128 grouped weight-gradient launch, 1: for the 64-tile group only (at most 4 problems), 0: neither."""
    return int(_lib.load().focal_linear_bwd_weight_group_kind(dtype_code, M, N, K))


def _dw_problems(items, exclusive):
    arr = (_lib.DwProblem * len(items))()
    for i, (dy, x, dw, db) in enumerate(items):
        _need_cuda(dy, x, dw, db)
        arr[i] = _lib.DwProblem(_p(dy), _p(x), _p(dw), _p(db), dy.shape[0], dy.shape[1], x.shape[1], 1 if exclusive else 0)
    return arr


DW_GROUP_MAX_PROBLEMS = 20  # FOCAL_DW_GROUP_MAX_PROBLEMS (include/focal_hip.h)


def linear_bwd_weight_group(dtype_code, items, exclusive=True):
    """items: [(dy [M, N], x [M, K], dw [N, K] fp32, dbias [N] fp32 | None)] -- the weight gradients of several linear layers as ONE
    launch (focal_linear_bwd_weight_group).  exclusive: nothing else adds to these dw while the launch runs."""
    arr = _dw_problems(items, exclusive)
    check(_lib.load().focal_linear_bwd_weight_group(dtype_code, len(items), arr, _stream()))


DW_TAIL_MAX = 8


def linear_bwd_weight_group_f32(compute_code, items, workgroups=0):
    """items: [(dy [M, N] fp32, x [M, K] fp32, dw [N, K] fp32, dbias [N] fp32 | None)], at most 8 -- weight gradients whose operands are fp32
    tensors (DeepSense's GRU) as ONE launch (focal_linear_bwd_weight_group_f32); `compute_code`: the matrix cores' operand type."""
    for dy, x, dw, db in items:
        if dy.dtype != torch.float32 or x.dtype != torch.float32 or dy.dim() != 2 or x.dim() != 2 or not (dy.is_contiguous() and x.is_contiguous()):
            raise _lib.FocalHipError("linear_bwd_weight_group_f32: dy and x must be contiguous fp32 matrices")
    check(_lib.load().focal_linear_bwd_weight_group_f32(compute_code, len(items), _dw_problems(items, False), workgroups, _stream()))


def linear_bwd_weight_group_workgroups(dtype_code, items, exclusive=True):
    return _lib.load().focal_linear_bwd_weight_group_workgroups(dtype_code, len(items), _dw_problems(items, exclusive))


def linear(x, w, bias=None, *, compute, y_dtype=None, resid=None, act_in=ACT_NONE, epilogue=EPI_NONE, splits=1,
           out_drop=None, act_grad=None):
    """Convenience forward: allocates y.  `compute` is the matrix-core operand dtype (torch.float32 / bfloat16)."""
    _need_cuda(x, w, bias, resid)
    M, K = x.shape
    N = w.shape[0]
    y_dtype = y_dtype or compute
    d = linear_desc(code(compute), M, N, K, code(x.dtype), code(y_dtype), act_in, epilogue, splits, out_drop)
    y = (torch.zeros if splits > 1 else torch.empty)(M, N, dtype=y_dtype, device=x.device)
    linear_fwd(d, x, w, bias, resid, y, act_grad)
    return y, d


# ------------------------------------------------------------------------------------------------ fused MLP branch
def mlp_supported(dtype, Cc, hidden):
    return bool(_lib.load().focal_mlp_supported(code(dtype), Cc, hidden))


def mlp_desc(dtype_code, M, Cc, hidden, drop_hidden=None, drop_out=None, ln_eps=1e-5):
    return MlpDesc(dtype_code, M, Cc, hidden, drop_hidden or NO_DROP, drop_out or NO_DROP, ln_eps)


def mlp_mask_bits(d, device):
    """The hidden-dropout keep bits of one fused-MLP call (int32 [M, 8]; None when the descriptor has no hidden dropout): filled by
    mlp_fwd, read by mlp_bwd -- together with `a` the saved state of the branch."""
    if d.drop_hidden.p_elem <= 0.0:
        return None
    return torch.empty(d.M, 8, dtype=torch.int32, device=device)


def mlp_fwd(d, a, resid, w1, b1, w2, b2, y, next_ln=None, mask_bits=None):
    """y = resid + drop_out(drop_hidden(gelu(a w1^T + b1)) w2^T + b2) (focal_mlp_fwd); next_ln = (gamma, beta) also returns
    (LayerNorm(y) in a's dtype, stats) of the LayerNorm that reads y next.  mask_bits (ops.mlp_mask_bits) receives the hidden
    dropout's keep bits for mlp_bwd."""
    _need_cuda(a, resid, w1, b1, w2, b2, y, mask_bits)
    if next_ln is None:
        check(_lib.load().focal_mlp_fwd(C.byref(d), _p(a), _p(resid), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), None, None, None, None,
                                            _p(mask_bits), _stream()))
        return None
    y_ln = torch.empty(d.M, d.C, dtype=a.dtype, device=a.device)
    stats = torch.empty(d.M, 2, dtype=torch.float32, device=a.device)
    check(_lib.load().focal_mlp_fwd(C.byref(d), _p(a), _p(resid), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), _p(next_ln[0]),
                                        _p(next_ln[1]), _p(y_ln), _p(stats), _p(mask_bits), _stream()))
    return y_ln, stats


def mlp_proj_supported(dtype, Cc, hidden):
    return bool(_lib.load().focal_mlp_proj_supported(code(dtype), Cc, hidden))


def mlp_proj_fwd(d, o, x, wp, bp, drop_proj, g2, bt2, x_mid, w1, b1, w2, b2, y, next_ln=None, mask_bits=None):
    """The attention branch's tail and the MLP branch of a 64-channel block in one launch (focal_mlp_proj_fwd): x_mid = x + drop_proj(o wp^T +
    bp) (written to `x_mid`), a2 = LayerNorm(x_mid; g2, bt2), then mlp_fwd.  Returns (a2, st2) and, with next_ln, (y_ln, stats) of the
    LayerNorm that reads y next."""
    _need_cuda(o, x, wp, bp, g2, bt2, x_mid, w1, b1, w2, b2, y, mask_bits)
    a2 = torch.empty(d.M, d.C, dtype=o.dtype, device=o.device)
    st2 = torch.empty(d.M, 2, dtype=torch.float32, device=o.device)
    mask = drop_proj or NO_DROP
    if next_ln is None:
        check(_lib.load().focal_mlp_proj_fwd(C.byref(d), _p(o), _p(x), _p(wp), _p(bp), C.byref(mask), _p(g2), _p(bt2), _p(x_mid), _p(a2), _p(st2), _p(w1), _p(b1),
                                             _p(w2), _p(b2), _p(y), None, None, None, None, _p(mask_bits), _stream()))
        return (a2, st2), None
    y_ln = torch.empty(d.M, d.C, dtype=o.dtype, device=o.device)
    stats = torch.empty(d.M, 2, dtype=torch.float32, device=o.device)
    check(_lib.load().focal_mlp_proj_fwd(C.byref(d), _p(o), _p(x), _p(wp), _p(bp), C.byref(mask), _p(g2), _p(bt2), _p(x_mid), _p(a2), _p(st2), _p(w1), _p(b1),
                                         _p(w2), _p(b2), _p(y), _p(next_ln[0]), _p(next_ln[1]), _p(y_ln), _p(stats), _p(mask_bits), _stream()))
    return (a2, st2), (y_ln, stats)


def mlp_wide_supported(dtype, Cc, hidden):
    return bool(_lib.load().focal_mlp_wide_supported(code(dtype), Cc, hidden))


def mlp_wide_fwd(d, a, resid, w1, b1, w2, b2, y, h, hg, next_ln=None):
    """The MLP branch at 128 / 256 channels as one launch (focal_mlp_wide_fwd): h, hg = drop_hidden(gelu(a w1^T + b1)) and its derivative
    x mask ([M, hidden], written for the backward pass), y = resid + drop_out(h w2^T + b2); next_ln = (gamma, beta) (128 channels) also
    returns (LayerNorm(y) in a's dtype, stats).  Bit-identical to linear_fwd(GELU) + linear_fwd / linear_resid_ln_fwd."""
    _need_cuda(a, resid, w1, b1, w2, b2, y, h, hg)
    assert h.shape == (d.M, d.hidden) and hg.shape == h.shape and h.dtype == a.dtype and hg.dtype == a.dtype and h.is_contiguous() and hg.is_contiguous()
    if next_ln is None:
        check(_lib.load().focal_mlp_wide_fwd(C.byref(d), _p(a), _p(resid), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), _p(h), _p(hg), None, None, None, None, _stream()))
        return None
    y_ln = torch.empty(d.M, d.C, dtype=a.dtype, device=a.device)
    stats = torch.empty(d.M, 2, dtype=torch.float32, device=a.device)
    check(_lib.load().focal_mlp_wide_fwd(C.byref(d), _p(a), _p(resid), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), _p(h), _p(hg), _p(next_ln[0]), _p(next_ln[1]),
                                         _p(y_ln), _p(stats), _stream()))
    return y_ln, stats


def mlp_wide_proj_supported(dtype, Cc, hidden):
    return bool(_lib.load().focal_mlp_wide_proj_supported(code(dtype), Cc, hidden))


def mlp_wide_proj_fwd(d, o, x, wp, bp, drop_proj, g2, bt2, x_mid, w1, b1, w2, b2, y, h, hg, next_ln=None):
    """mlp_proj_fwd's form at 128 / 256 channels (focal_mlp_wide_proj_fwd): x_mid = x + drop_proj(o wp^T + bp) (written to `x_mid`),
    a2 = LayerNorm(x_mid; g2, bt2), then mlp_wide_fwd.  Returns (a2, st2), (y_ln, stats) | None."""
    _need_cuda(o, x, wp, bp, g2, bt2, x_mid, w1, b1, w2, b2, y, h, hg)
    a2 = torch.empty(d.M, d.C, dtype=o.dtype, device=o.device)
    st2 = torch.empty(d.M, 2, dtype=torch.float32, device=o.device)
    mask = drop_proj or NO_DROP
    if next_ln is None:
        check(_lib.load().focal_mlp_wide_proj_fwd(C.byref(d), _p(o), _p(x), _p(wp), _p(bp), C.byref(mask), _p(g2), _p(bt2), _p(x_mid), _p(a2), _p(st2), _p(w1), _p(b1),
                                                  _p(w2), _p(b2), _p(y), _p(h), _p(hg), None, None, None, None, _stream()))
        return (a2, st2), None
    y_ln = torch.empty(d.M, d.C, dtype=o.dtype, device=o.device)
    stats = torch.empty(d.M, 2, dtype=torch.float32, device=o.device)
    check(_lib.load().focal_mlp_wide_proj_fwd(C.byref(d), _p(o), _p(x), _p(wp), _p(bp), C.byref(mask), _p(g2), _p(bt2), _p(x_mid), _p(a2), _p(st2), _p(w1), _p(b1),
                                              _p(w2), _p(b2), _p(y), _p(h), _p(hg), _p(next_ln[0]), _p(next_ln[1]), _p(y_ln), _p(stats), _stream()))
    return (a2, st2), (y_ln, stats)


def mlp_wide_bwd_supported(dtype, Cc, hidden):
    return bool(_lib.load().focal_mlp_wide_bwd_supported(code(dtype), Cc, hidden))


def mlp_wide_bwd_data(d, gm, hg, w1, w2, du, dc=None, ln=None):
    """du = (gm w2) x hg and dc = du w1 in one launch (focal_mlp_wide_bwd_data).  dc: the [M, C] output; or ln = dict(x=, stats=, gamma=, g=,
    g_masked=, mask=, dgamma=, dbeta=) (128 channels): norm2's backward finished on the row as linear_bwd_data_ln does."""
    _need_cuda(gm, hg, w1, w2, du, dc)
    assert du.shape == (d.M, d.hidden) and du.dtype == gm.dtype and du.is_contiguous() and hg.shape == du.shape
    if ln is None:
        assert dc is not None and dc.shape == (d.M, d.C) and dc.dtype == gm.dtype and dc.is_contiguous()
        check(_lib.load().focal_mlp_wide_bwd_data(C.byref(d), _p(gm), _p(hg), _p(w1), _p(w2), _p(du), _p(dc), None, None, None, None, None, None, None, None, _stream()))
        return
    mask = ln.get("mask") or NO_DROP
    check(_lib.load().focal_mlp_wide_bwd_data(C.byref(d), _p(gm), _p(hg), _p(w1), _p(w2), _p(du), None, _p(ln["x"]), _p(ln["stats"]), _p(ln["gamma"]), _p(ln["g"]),
                                              _p(ln.get("g_masked")), C.byref(mask), _p(ln["dgamma"]), _p(ln["dbeta"]), _stream()))


def mlp_bwd_partials_floats(d):
    return int(_lib.load().focal_mlp_bwd_partials_floats(C.byref(d)))


def mlp_bwd_partials(d, device):
    """The workspace focal_mlp_bwd sums its workgroups' weight-gradient images through (instead of fp32 atomics): fp32 [workgroups * 32768]."""
    return torch.empty(int(_lib.load().focal_mlp_bwd_partials_floats(C.byref(d))), dtype=torch.float32, device=device)


def mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2, ln=None, mask_bits=None, partials=None):
    """focal_mlp_bwd.  mask_bits: what mlp_fwd filled (required when the descriptor has hidden dropout).  ln = dict(x=, stats=, gamma=,
    g=, gm_next=, next_mask=, dgamma=, dbeta=) fuses the backward of the LayerNorm that produced `a` (norm2) into the kernel: g += dLN,
    gm_next = dtype(g x next_mask), dgamma / dbeta accumulate, and `da` is neither needed nor written (pass None) -- what the Swin engine
    always does.  partials (mlp_bwd_partials): the weight gradients leave through a workspace + a reduce launch instead of atomics."""
    _need_cuda(gm, a, w1, b1, w2, da, dw1, db1, dw2, db2, mask_bits, partials)
    if ln is None:
        check(_lib.load().focal_mlp_bwd(C.byref(d), _p(gm), _p(a), _p(w1), _p(b1), _p(w2), _p(da), _p(dw1), _p(db1), _p(dw2), _p(db2),
                                        None, None, None, None, None, None, None, None, _p(mask_bits), _p(partials), _stream()))
        return
    mask = ln.get("next_mask") or NO_DROP
    check(_lib.load().focal_mlp_bwd(C.byref(d), _p(gm), _p(a), _p(w1), _p(b1), _p(w2), _p(da), _p(dw1), _p(db1), _p(dw2), _p(db2),
                                    _p(ln["x"]), _p(ln["stats"]), _p(ln["gamma"]), _p(ln["g"]), _p(ln.get("gm_next")), C.byref(mask),
                                    _p(ln["dgamma"]), _p(ln["dbeta"]), _p(mask_bits), _p(partials), _stream()))


# ------------------------------------------------------------------------------------------------ row 10
def attn_desc(dtype_code, B, H, W, Cc, heads, wh, ww, sh, sw, p_attn=0.0, rng=None, stream=0):
    return AttnDesc(dtype_code, B, H, W, Cc, heads, wh, ww, sh, sw, p_attn, _p(rng), stream)


def window_attn_fwd(d, qkv, bias_table, out):
    check(_lib.load().focal_window_attn_fwd(C.byref(d), _p(qkv), _p(bias_table), _p(out), _stream()))


def window_attn_bwd(d, qkv, bias_table, dout, dqkv, dbias_table):
    check(_lib.load().focal_window_attn_bwd(C.byref(d), _p(qkv), _p(bias_table), _p(dout), _p(dqkv), _p(dbias_table),
                                            _stream()))


def attn_qkv_supported(dtype_code, Cc, heads, window_tokens):
    return bool(_lib.load().focal_window_attn_qkv_supported(dtype_code, Cc, heads, window_tokens))


def window_attn_qkv_fwd(d, a1, wqkv, bqkv, bias_table, out):
    """W-MSA with the qkv Linear folded in (64-channel blocks): a1 [M, C] -> out [M, C]; no [M, 3C] tensor."""
    _need_cuda(a1, wqkv, bqkv, bias_table, out)
    check(_lib.load().focal_window_attn_qkv_fwd(C.byref(d), _p(a1), _p(wqkv), _p(bqkv), _p(bias_table), _p(out), _stream()))


def window_attn_qkv_bwd(d, a1, wqkv, bqkv, bias_table, dout, dqkv, dbias_table, wproj=None):
    """wproj given: `dout` is the gradient w.r.t. the proj Linear's OUTPUT (masked, operand dtype) and the kernel forms dout . wproj itself."""
    _need_cuda(a1, wqkv, bqkv, bias_table, dout, dqkv, dbias_table, wproj)
    check(_lib.load().focal_window_attn_qkv_bwd(C.byref(d), _p(a1), _p(wqkv), _p(bqkv), _p(bias_table), _p(dout), _p(wproj), _p(dqkv),
                                                _p(dbias_table), _stream()))


# ------------------------------------------------------------------------------------------------ rows 11-13
_LOSS_WS = {}


def _loss_setup(feats1, feats2, temperature, margin, weights, seq, no_private):
    feats = list(feats1) + list(feats2)
    _need_cuda(*feats)
    M = len(feats1)
    B, dim = feats[0].shape
    dev = feats[0].device
    d = LossDesc(M, B, dim, seq, temperature, margin, weights[0], weights[1], weights[2], weights[3], int(no_private))
    lib = _lib.load()
    need = lib.focal_loss_head_workspace(C.byref(d))
    if need == 0:
        raise _lib.FocalHipError(f"loss_head: {lib.focal_last_error().decode()}")
    key = (dev, torch.cuda.current_stream().cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _LOSS_WS[key] = ws
    # one allocation [2M gradients | terms(5) + pad]: the library zeroes it with one memset, the autograd node scales it with one launch
    n = B * dim
    flat = torch.empty(2 * M * n + 8, dtype=torch.float32, device=dev)
    # ABI order is view-major (index v * M + m); memory order is modality-major, so a modality's two view gradients are the
    # two halves of one [2B, dim] block (what a backbone that ran both views as one batch wants back, without a concatenation)
    grads = [flat[(m * 2 + v) * n:(m * 2 + v + 1) * n].view(B, dim) for v in range(2) for m in range(M)]
    terms = flat[2 * M * n:2 * M * n + 5]
    fa = (C.c_void_p * (2 * M))(*[_p(f) for f in feats])
    ga = (C.c_void_p * (2 * M))(*[_p(g) for g in grads])
    return lib, d, M, n, feats, flat, grads, terms, fa, ga, ws


def loss_head(feats1, feats2, temperature, margin, weights, seq=4, no_private=False, return_flat=False):
    """feats{1,2}: lists (modality order) of fp32 [B, dim].  Returns (terms[5] device tensor, grads1, grads2)."""
    lib, d, M, n, feats, flat, grads, terms, fa, ga, ws = _loss_setup(feats1, feats2, temperature, margin, weights, seq, no_private)
    check(lib.focal_loss_head(C.byref(d), fa, _p(terms), ga, _p(ws), ws.numel(), _stream()))
    if return_flat:  # the gradients as ONE tensor (they are views of it)
        return terms, grads[:M], grads[M:], flat[:2 * M * n]
    return terms, grads[:M], grads[M:]


class ShardedLossHead:
    """The loss head of `world` data-parallel ranks, each over the rows of its own samples (focal_loss_head_shard_a / _b).
    `phase_a(...)` fills the persistent `send` buffer [n]; the caller all-gathers it into `chunks` [world, n];
    `phase_b()` returns what `loss_head(..., return_flat=True)` returns; only the rank's own rows of the gradients are non-zero."""

    _BUF = {}

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.state = None

    def exchange_buffers(self, d, dev):
        n = _lib.load().focal_loss_head_exchange_floats(C.byref(d), self.world)
        if n == 0:
            raise _lib.FocalHipError(f"loss_head (sharded): {_lib.load().focal_last_error().decode()}")
        key = (dev, self.world, n)
        if key not in self._BUF:  # address-stable: captured graph segments write / read them, the collective between them is eager
            self._BUF[key] = (torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(self.world, n, dtype=torch.float32, device=dev))
        return self._BUF[key]

    def phase_a(self, feats1, feats2, temperature, margin, weights, seq=4, no_private=False):
        lib, d, M, n, feats, flat, grads, terms, fa, ga, ws = _loss_setup(feats1, feats2, temperature, margin, weights, seq, no_private)
        self.send, self.chunks = self.exchange_buffers(d, feats[0].device)
        check(lib.focal_loss_head_shard_a(C.byref(d), self.rank, self.world, fa, _p(terms), ga, _p(self.send), _p(ws), ws.numel(), _stream()))
        self.state = (lib, d, M, n, feats, flat, grads, terms, fa, ga, ws)
        return self.send, self.chunks

    def phase_b(self):
        lib, d, M, n, feats, flat, grads, terms, fa, ga, ws = self.state
        self.state = None
        check(lib.focal_loss_head_shard_b(C.byref(d), self.rank, self.world, fa, _p(terms), ga, _p(self.chunks), _p(ws), ws.numel(), _stream()))
        return terms, grads[:M], grads[M:], flat[:2 * M * n]


# ------------------------------------------------------------------------------------------------ rows 5-6 (DeepSense)
def conv_in_desc(B, cin, I, S_in, S_out, k, stride, pad_left, Cc):
    return ConvInDesc(B, cin, I, S_in, S_out, k, stride, pad_left, Cc)


def conv_in_fwd(d, x, w, bias):
    z = torch.empty(d.B * d.I * d.S_out, d.C, dtype=torch.float32, device=x.device)
    check(_lib.load().focal_conv_in_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(z), _stream()))
    return z


def conv_in_bwd_weight(d, x, dz, dw, dbias):
    check(_lib.load().focal_conv_in_bwd_weight(C.byref(d), _p(x), _p(dz), code(dz.dtype), _p(dw), _p(dbias), _stream()))


def permute_pack(src, A, Bd, Cd, dtype):
    """dst[a][c][b] = src[a][b][c], cast to `dtype`."""
    dst = torch.empty(A, Cd, Bd, dtype=dtype, device=src.device)
    check(_lib.load().focal_permute_pack(A, Bd, Cd, _p(src), _p(dst), code(dtype), _stream()))
    return dst


def permute_unpack_add(src, dst, A, Bd, Cd):
    """dst[a][b][c] += src[a][c][b] (fp32)."""
    check(_lib.load().focal_permute_unpack_add(A, Bd, Cd, _p(src), _p(dst), _stream()))


PACK_PERMUTE, PACK_CONV_BWD, PACK_FRAG, PACK_FRAG_T = 0, 1, 2, 3
PACK_MAX = 24


def pack_multi(entries, dtype):
    """entries: [(src fp32, dst `dtype`, A, B, C, kind)] -- every re-ordering of an encoder's weights in one launch."""
    arr = (_lib.PackEntry * len(entries))()
    for i, (src, dst, A, Bd, Cd, kind) in enumerate(entries):
        _need_cuda(src, dst)
        arr[i] = _lib.PackEntry(_p(src), _p(dst), A, Bd, Cd, kind)
    check(_lib.load().focal_pack_multi(code(dtype), len(entries), arr, _stream()))


def unpack_add_multi(entries):
    """entries: [(src packed fp32 [A][C][B], dst fp32 [A][B][C], A, B, C)]: dst += unpacked src, one launch."""
    arr = (_lib.PackEntry * len(entries))()
    for i, (src, dst, A, Bd, Cd) in enumerate(entries):
        _need_cuda(src, dst)
        arr[i] = _lib.PackEntry(_p(src), _p(dst), A, Bd, Cd, PACK_PERMUTE)
    check(_lib.load().focal_unpack_add_multi(len(entries), arr, _stream()))


def conv_desc(dtype_code, rows, S, C_in, C_out, k, dw_workgroups=0):
    return ConvDesc(dtype_code, rows, S, C_in, C_out, k, dw_workgroups)


def conv_pack_bwd(d, w, dtype):
    dst = torch.empty(d.C_in, d.k, d.C_out, dtype=dtype, device=w.device)
    check(_lib.load().focal_conv_pack_bwd(C.byref(d), _p(w), _p(dst), _stream()))
    return dst


def conv_fwd(d, x, w_fwd, bias):
    z = torch.empty(d.rows, d.C_out, dtype=torch.float32, device=x.device)
    check(_lib.load().focal_conv_fwd(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(z), _stream()))
    return z


def conv_fwd_bn_sums_supported(d, d_bn, x, w_fwd):
    return bool(_lib.load().focal_conv_fwd_bn_sums_supported(C.byref(d), C.byref(d_bn), _p(x), _p(w_fwd)))


def conv_fwd_bn_sums(d, x, w_fwd, bias, d_bn):
    """focal_conv_fwd_bn in its sums-only form (round 6): -> (z, sums); bn_act_fwd_sums finishes the statistics."""
    dev = x.device
    z = torch.empty(d.rows, d.C_out, dtype=torch.float32, device=dev)
    n = _bn_groups(d_bn) * (_lib.BN_STAT_SLOTS * 2 * d_bn.C + 1)
    scratch = pool_zeros(n, dev)
    if scratch is None:
        scratch = torch.zeros(n, dtype=torch.float32, device=dev)
    check(_lib.load().focal_conv_fwd_bn(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(z), C.byref(d_bn), _p(scratch), None, None, None, _stream()))
    return z, scratch


def bn_act_fwd_sums(d, z, sums, running_mean, running_var, gamma, beta, resid, cast_dtype=None):
    """bn_act_fwd on a convolution's per-channel sums (conv_fwd_bn_sums): -> (y, y_cast, mean_rstd)."""
    y = torch.empty_like(z)
    ya = torch.empty(z.shape, dtype=cast_dtype, device=z.device) if cast_dtype not in (None, torch.float32) else None
    mean_rstd = torch.empty(_bn_groups(d) * 2 * d.C, dtype=torch.float32, device=z.device)
    check(_lib.load().focal_bn_act_fwd_sums(C.byref(d), _p(z), _p(sums), _p(mean_rstd), _p(running_mean), _p(running_var), _p(gamma), _p(beta),
                                            _p(resid), _p(y), _p(ya), _stream()))
    return y, (ya if ya is not None else y), mean_rstd


def conv_fwd_bn(d, x, w_fwd, bias, d_bn, running_mean, running_var):
    """conv_fwd + the training-mode statistics of the BatchNorm behind it in one launch (focal_conv_fwd_bn): -> (z, mean_rstd).  One rank /
    no sync_bn (the global-batch form all-reduces the sums between two kernels: conv_fwd + bn_stats)."""
    dev = x.device
    z = torch.empty(d.rows, d.C_out, dtype=torch.float32, device=dev)
    G = _bn_groups(d_bn)
    n = G * (_lib.BN_STAT_SLOTS * 2 * d_bn.C + 1)
    scratch = pool_zeros(n, dev)
    if scratch is None:
        scratch = torch.zeros(n, dtype=torch.float32, device=dev)
    mean_rstd = torch.empty(G * 2 * d_bn.C, dtype=torch.float32, device=dev)
    check(_lib.load().focal_conv_fwd_bn(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(z), C.byref(d_bn), _p(scratch), _p(mean_rstd),
                                        _p(running_mean), _p(running_var), _stream()))
    return z, mean_rstd


def conv_bwd_data(d, dz, w_bwd, g_in, g_out):
    check(_lib.load().focal_conv_bwd_data(C.byref(d), _p(dz), _p(w_bwd), _p(g_in), _p(g_out), _stream()))


def conv_bwd_weight(d, dz, x, dw_packed, dbias):
    check(_lib.load().focal_conv_bwd_weight(C.byref(d), _p(dz), _p(x), _p(dw_packed), _p(dbias), _stream()))


def bn_desc(dtype_code, rows, Cc, rows_per_sample, p_drop=0.0, rng=None, stream=0, eps=1e-5, momentum=0.1, stat_rows=0, groups=1):
    """groups > 1: the rows are `groups` equal consecutive ranges with batch statistics of their own (the two views of a FOCAL step as
    one batch): mean_rstd [groups, 2C], the running buffers handed to bn_stats / conv_fwd_bn are per-group sinks [groups, C]."""
    return BNDesc(dtype_code, rows, Cc, rows_per_sample, eps, momentum, p_drop, _p(rng), stream, stat_rows, groups)


def _bn_groups(d):
    return d.groups if d.groups > 1 else 1


def _sync_world():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def bn_stats(d, z, running_mean, running_var, training, sync=False):
    """mean / rstd of z [rows, C].  sync=True under torch.distributed: the 2C per-channel sums are all-reduced between the
    partial and the finalize kernels, so every rank normalises with the statistics of the GLOBAL batch (equal shards)."""
    dev = running_mean.device
    G = _bn_groups(d)
    scratch = pool_zeros(G * (2 * d.C + 1), dev) if training else None  # {sum, sum of squares} + the statistics kernel's arrival counter
    pre = _lib.BN_SCRATCH_ZEROED if scratch is not None else 0
    if scratch is None:
        scratch = torch.empty(G * (2 * d.C + 1), dtype=torch.float32, device=dev)
    mean_rstd = torch.empty(G * 2 * d.C, dtype=torch.float32, device=dev)
    lib = _lib.load()
    args = (_p(z), _p(scratch), _p(mean_rstd), _p(running_mean), _p(running_var))
    world = _sync_world() if (sync and training) else 1
    if world > 1:
        import torch.distributed as dist
        check(lib.focal_bn_stats(C.byref(d), *args, _lib.BN_PARTIAL | pre, _stream()))
        dist.all_reduce(scratch[:2 * d.C], op=dist.ReduceOp.SUM)
        d.stat_rows = d.rows * world
        check(lib.focal_bn_stats(C.byref(d), *args, _lib.BN_FINALIZE, _stream()))
    else:
        check(lib.focal_bn_stats(C.byref(d), *args, (_lib.BN_TRAIN | pre) if training else _lib.BN_EVAL, _stream()))
    return mean_rstd


def bn_running_combine(running, view1, view2, momentum=0.1):
    """running[i] <- (1 - m) ((1 - m) running[i] + m view1[i]) + m view2[i] for n same-sized fp32 buffers, one launch."""
    _need_cuda(*running, *view1, *view2)
    check(_lib.load().focal_bn_running_combine(len(running), _parr(running), _parr(view1), _parr(view2), running[0].numel(), momentum, _stream()))


def bn_act_fwd(d, z, mean_rstd, gamma, beta, resid, cast_dtype=None):
    y = torch.empty_like(z)
    ya = torch.empty(z.shape, dtype=cast_dtype, device=z.device) if cast_dtype not in (None, torch.float32) else None
    check(_lib.load().focal_bn_act_fwd(C.byref(d), _p(z), _p(mean_rstd), _p(gamma), _p(beta), _p(resid), _p(y), _p(ya), _stream()))
    return y, (ya if ya is not None else y)


def bn_act_bwd(d, z, g, mean_rstd, gamma, beta, dgamma, dbeta, out_dtype, sync=False):
    G = _bn_groups(d)
    scratch = pool_zeros(G * (2 * d.C + 1), z.device)
    pre = _lib.BN_SCRATCH_ZEROED if scratch is not None else 0
    if scratch is None:
        scratch = torch.empty(G * (2 * d.C + 1), dtype=torch.float32, device=z.device)
    dz = torch.empty(z.shape, dtype=out_dtype, device=z.device)
    lib = _lib.load()
    args = (_p(z), _p(g), _p(mean_rstd), _p(gamma), _p(beta), _p(scratch), _p(dz), _p(dgamma), _p(dbeta))
    world = _sync_world() if sync else 1
    if world > 1:
        import torch.distributed as dist
        check(lib.focal_bn_act_bwd(C.byref(d), *args, _lib.BN_PARTIAL | pre, _stream()))
        dist.all_reduce(scratch[:2 * d.C], op=dist.ReduceOp.SUM)
        d.stat_rows = d.rows * world
        check(lib.focal_bn_act_bwd(C.byref(d), *args, _lib.BN_FINALIZE, _stream()))
    else:
        check(lib.focal_bn_act_bwd(C.byref(d), *args, _lib.BN_TRAIN | pre, _stream()))
    return dz


def gru_gate_fwd(d, t, dir_off, gi, gh, h_prev, h_new, out, save):
    check(_lib.load().focal_gru_gate_fwd(C.byref(d), t, dir_off, _p(gi), _p(gh), _p(h_prev), _p(h_new), _p(out), _p(save), _stream()))


def gru_gate_bwd(d, t, dir_off, dout, ld_b, ld_t, scale, dh_rec, dhz_in, save, h_prev, dgi, dgh, dhz_out):
    check(_lib.load().focal_gru_gate_bwd(C.byref(d), t, dir_off, _p(dout), ld_b, ld_t, scale, _p(dh_rec), _p(dhz_in), _p(save),
                                         _p(h_prev), _p(dgi), _p(dgh), _p(dhz_out), _stream()))


def _parr(tensors):
    return (C.c_void_p * len(tensors))(*[_p(t) for t in tensors])


def gru_seq_fwd(d, gi, whh, bhh, hs, save, out):
    """Whole-sequence GRU layer, one launch for len(gi) directions (lists of per-direction tensors)."""
    check(_lib.load().focal_gru_seq_fwd(C.byref(d), len(gi), _parr(gi), _parr(whh), _parr(bhh), _parr(hs), _parr(save), _p(out), _stream()))


def gru_seq_bwd(d, dout, ld_b, ld_t, scale, whh_t, hs, save, dgi, dgh):
    check(_lib.load().focal_gru_seq_bwd(C.byref(d), len(hs), _p(dout), ld_b, ld_t, scale, _parr(whh_t), _parr(hs), _parr(save), _parr(dgi),
                                        _parr(dgh), _stream()))


def fusion_attn_fwd(B, M, E, heads, q, kv, out, probs, weights, rng, stream_id, p_drop):
    check(_lib.load().focal_fusion_attn_fwd(B, M, E, heads, _p(q), _p(kv), _p(out), _p(probs), _p(weights), _p(rng), stream_id, p_drop, _stream()))


def fusion_attn_bwd(B, M, E, heads, q, kv, probs, weights, dout, dq, dkv):
    check(_lib.load().focal_fusion_attn_bwd(B, M, E, heads, _p(q), _p(kv), _p(probs), _p(weights), _p(dout), _p(dq), _p(dkv), _stream()))


def small_linear_fwd(x, w, bias):
    B, K = x.shape
    y = torch.empty(B, w.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().focal_small_linear_fwd(B, w.shape[0], K, _p(x), _p(w), _p(bias), _p(y), _stream()))
    return y


def small_linear_bwd(dy, x, w, dw, dbias, need_dx=True):
    B, K = x.shape
    dx = torch.empty_like(x) if need_dx else None
    check(_lib.load().focal_small_linear_bwd(B, w.shape[0], K, _p(dy), _p(x), _p(w), _p(dw), _p(dbias), _p(dx), _stream()))
    return dx


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() (mean): returns (loss [1] fp32, dlogits [B, C])."""
    _need_cuda(logits, labels)
    B, Cn = logits.shape
    logits = logits.float().contiguous()
    labels = labels.long().contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(_lib.load().focal_cross_entropy(B, Cn, _p(logits), _p(labels), _p(loss), _p(dlogits), _stream()))
    return loss, dlogits


def mean_time(x, B, T, D):
    y = torch.empty(B, D, dtype=torch.float32, device=x.device)
    check(_lib.load().focal_mean_time(B, T, D, _p(x), _p(y), _stream()))
    return y


def dropout(x, rng, stream_id, p):
    y = torch.empty_like(x)
    check(_lib.load().focal_dropout(x.numel(), _p(x), _p(y), _p(rng), stream_id, p, _stream()))
    return y


def mul_(y, a):
    """y *= a (fp32, element-wise; focal_mul)."""
    _need_cuda(y, a)
    check(_lib.load().focal_mul(y.numel(), _p(a), _p(y), _stream()))
    return y


def axpy(a, x, y):
    check(_lib.load().focal_axpy(x.numel(), a, _p(x), _p(y), _stream()))
