"""Fused Swin MLP branch (focal_mlp_fwd / focal_mlp_bwd, reference: models/SwinModules.py:18-34 + :339-341) through the C ABI:
against torch fp32 autograd of the same expression on the bf16-rounded operands (dropout off), and -- with every dropout /
drop-path mask switched on -- against the unfused HIP path (fc1 GELU epilogue -> fc2 residual epilogue; dW / dX GEMMs), which
draws the same masks from the same (seed, stream, element) hash."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def _operands(M, C=64, seed=0):
    a = rnd(M, C, seed=seed + 1, dtype=BF)
    w1, b1 = rnd(4 * C, C, scale=C ** -0.5, seed=seed + 2, dtype=BF), rnd(4 * C, scale=0.3, seed=seed + 3)
    w2, b2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=seed + 4, dtype=BF), rnd(C, scale=0.3, seed=seed + 5)
    r = rnd(M, C, seed=seed + 6)
    return a, w1, b1, w2, b2, r


@pytest.fixture(scope="module")
def ops():
    from focal_amd import ops as o
    return o


@pytest.mark.parametrize("M", [16, 100, 128, 4608, 36864 + 48])
def test_fused_mlp_forward_and_next_layernorm_match_torch(ops, M):
    C = 64
    a, w1, b1, w2, b2, r = _operands(M)
    assert ops.mlp_supported(BF, C, 4 * C) and not ops.mlp_supported(torch.float32, C, 4 * C) and not ops.mlp_supported(BF, 128, 512)
    d = ops.mlp_desc(ops.code(BF), M, C, 4 * C)
    y = torch.empty(M, C, device=DEV)
    gamma, beta = rnd(C, seed=11) * 0.2 + 1.0, rnd(C, seed=12) * 0.1
    y_ln, stats = ops.mlp_fwd(d, a, r, w1, b1, w2, b2, y, next_ln=(gamma, beta))
    h = F.gelu(a.float() @ w1.float().t() + b1).to(BF).float()  # the hidden activation is a bf16 matrix-core operand
    ref = r + h @ w2.float().t() + b2
    assert rel_err(y, ref) < 3e-3
    assert (y - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
    ln = F.layer_norm(y, (C,), gamma, beta, 1e-5)
    assert rel_err(y_ln.float(), ln) < 5e-3  # bf16 output rounding
    assert torch.allclose(stats[:, 0], y.mean(1), atol=1e-5) and torch.allclose(stats[:, 1], (y.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    y2 = torch.empty_like(y)
    assert ops.mlp_fwd(d, a, r, w1, b1, w2, b2, y2) is None
    assert torch.equal(y, y2)


@pytest.mark.parametrize("M", [128, 1000, 4608 + 16, 73728])
def test_fused_mlp_backward_matches_torch_autograd(ops, M):
    C = 64
    a, w1, b1, w2, b2, r = _operands(M, seed=20)
    gm = rnd(M, C, scale=0.5, seed=31, dtype=BF)
    d = ops.mlp_desc(ops.code(BF), M, C, 4 * C)
    da = torch.empty(M, C, dtype=BF, device=DEV)
    dw1, db1 = torch.zeros(4 * C, C, device=DEV), torch.zeros(4 * C, device=DEV)
    dw2, db2 = torch.zeros(C, 4 * C, device=DEV), torch.zeros(C, device=DEV)
    ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2)
    at, w1t, b1t = a.float().requires_grad_(True), w1.float().requires_grad_(True), b1.clone().requires_grad_(True)
    w2t, b2t = w2.float().requires_grad_(True), b2.clone().requires_grad_(True)
    yt = F.gelu(at @ w1t.t() + b1t) @ w2t.t() + b2t
    yt.backward(gm.float())
    # bf16 rounding of h and du (both matrix-core operands) bounds the agreement
    assert rel_err(da.float(), at.grad) < 8e-3
    assert rel_err(dw2, w2t.grad) < 6e-3 and rel_err(dw1, w1t.grad) < 8e-3
    assert rel_err(db2, b2t.grad) < 1e-4 and rel_err(db1, b1t.grad) < 6e-3
    # accumulation semantics: a second call adds onto the first
    ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2)
    assert rel_err(dw2, 2 * w2t.grad) < 6e-3 and rel_err(db1, 2 * b1t.grad) < 6e-3


def test_fused_mlp_equals_unfused_hip_path_without_dropout(ops):
    """Same roundings, no masks: the fused kernels must reproduce the unfused chain (fc1 GELU epilogue -> fc2 residual epilogue;
    dW / dX GEMMs) -- forward output, and the backward's da / dW / db -- to accumulation-order noise.  DropPath / output dropout stay
    on: they use the element hash both paths share."""
    from focal_amd._lib import ACT_GELU, EPI_GELU, EPI_RESIDUAL
    M, C, L = 9216, 64, 576
    a, w1, b1, w2, b2, r = _operands(M, seed=40)
    rng = ops.new_rng_state(1234, DEV)
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    drop_h = ops.drop_desc(rng, 11, 0.0, 15, 0.0, L)
    drop_o = ops.drop_desc(rng, 12, 0.2, 16, 0.1, L)
    d1 = ops.linear_desc(cc, M, 4 * C, C, cc, cc, 0, EPI_GELU, out_drop=drop_h)
    h, hg = torch.empty(M, 4 * C, dtype=BF, device=DEV), torch.empty(M, 4 * C, dtype=BF, device=DEV)
    ops.linear_fwd(d1, a, w1, b1, None, h, hg)
    d2 = ops.linear_desc(cc, M, C, 4 * C, cc, f32, ACT_GELU, EPI_RESIDUAL, out_drop=drop_o)
    y_ref = torch.empty(M, C, device=DEV)
    ops.linear_fwd(d2, h, w2, b2, r, y_ref)
    d = ops.mlp_desc(cc, M, C, 4 * C, drop_h, drop_o)
    y = torch.empty(M, C, device=DEV)
    ops.mlp_fwd(d, a, r, w1, b1, w2, b2, y)
    assert ((y - r) == 0).float().mean().item() > 0.15       # output dropout really on
    assert rel_err(y, y_ref) < 3e-4, rel_err(y, y_ref)       # (a bf16 rounding of h may flip where the two erf evaluations differ by 1e-7)
    g = rnd(M, C, scale=0.5, seed=51)
    gm = ops.mask_cast(g, drop_o, BF)
    d2b = ops.linear_desc(cc, M, C, 4 * C, cc, cc, ACT_GELU)
    dw2_ref, db2_ref = torch.zeros(C, 4 * C, device=DEV), torch.zeros(C, device=DEV)
    ops.linear_bwd_weight(d2b, gm, h, dw2_ref, db2_ref)
    du = torch.empty_like(h)
    ops.linear_bwd_data(d2b, gm, w2, hg, du)
    dw1_ref, db1_ref = torch.zeros(4 * C, C, device=DEV), torch.zeros(4 * C, device=DEV)
    ops.linear_bwd_weight(d1, du, a, dw1_ref, db1_ref)
    da_ref = torch.empty(M, C, dtype=BF, device=DEV)
    ops.linear_bwd_data(d1, du, w1, None, da_ref)
    da = torch.empty(M, C, dtype=BF, device=DEV)
    dw1, db1 = torch.zeros(4 * C, C, device=DEV), torch.zeros(4 * C, device=DEV)
    dw2, db2 = torch.zeros(C, 4 * C, device=DEV), torch.zeros(C, device=DEV)
    ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2)
    assert rel_err(da.float(), da_ref.float()) < 4e-3       # two bf16 roundings of the same fp32 value may differ by one ulp
    assert rel_err(dw2, dw2_ref) < 3e-4 and rel_err(db2, db2_ref) < 2e-5
    assert rel_err(dw1, dw1_ref) < 6e-3 and rel_err(db1, db1_ref) < 6e-3   # the unfused chain multiplies by the bf16-ROUNDED saved derivative, the fused one by the fp32 value


def test_fused_mlp_hidden_dropout_rate_scale_and_forward_backward_consistency(ops):
    """The hidden-activation dropout of the fused kernels (a per-(row, lane group) xorshift stream, csrc/mlp.hpp): the mask is
    read out of the FORWARD by making fc2 a selector of 64 hidden units (y - resid = dropped h, b2 = 0, no output dropout), and out
    of the BACKWARD through db1 with a one-hot output gradient (db1[h] = du[r][h], zero exactly where the unit was dropped);
    they must be the same mask, drop a fraction p of the units and scale survivors by 1 / (1 - p); a new seed gives a new mask."""
    M, C, H, p = 128, 64, 256, 0.2
    a, w1, b1, _, _, _ = _operands(M, seed=60)
    r = torch.zeros(M, C, device=DEV)
    b2 = torch.zeros(C, device=DEV)
    rng = ops.new_rng_state(77, DEV)
    cc = ops.code(BF)
    d = ops.mlp_desc(cc, M, C, H, ops.drop_desc(rng, 21, p, 25, 0.0, 16), ops.drop_desc(rng, 22, 0.0, 26, 0.0, 16))
    d0 = ops.mlp_desc(cc, M, C, H)
    fwd_mask = torch.zeros(M, H, dtype=torch.bool, device=DEV)   # True = kept
    live_fwd = torch.zeros(M, H, dtype=torch.bool, device=DEV)   # units whose un-dropped activation is clearly non-zero
    bits = ops.mlp_mask_bits(d, DEV)                             # the keep bits the forward kernel saves for the backward kernel
    assert ops.mlp_mask_bits(d0, DEV) is None
    ratio = []
    for quarter in range(4):
        w2 = torch.zeros(C, H, device=DEV)
        w2[torch.arange(C), quarter * C + torch.arange(C)] = 1.0
        w2 = w2.to(BF)
        y, y0 = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
        ops.mlp_fwd(d, a, r, w1, b1, w2, b2, y, mask_bits=bits)
        ops.mlp_fwd(d0, a, r, w1, b1, w2, b2, y0)
        live = y0.abs() > 1e-3                                   # units whose un-dropped activation is clearly non-zero
        fwd_mask[:, quarter * C:(quarter + 1) * C] = (y != 0) | ~live
        live_fwd[:, quarter * C:(quarter + 1) * C] = live
        sel = live & (y != 0)
        ratio.append((y[sel] / y0[sel]).float())
    ratio = torch.cat(ratio)
    assert (ratio - 1.0 / (1.0 - p)).abs().max().item() < 2e-2   # survivors scaled by 1.25 (bf16 rounding of h)
    rate = 1.0 - fwd_mask.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    # the saved words: hidden unit 16 T + 4 g + e of a row = bit 4 (T % 8) + e of word 2 g + T / 8 (include/focal_hip.h)
    hid = torch.arange(H, device=DEV)
    T, gq, e = hid // 16, (hid % 16) // 4, hid % 4
    word = bits.to(torch.int64)[:, (2 * gq + T // 8)] & 0xFFFFFFFF
    saved = ((word >> (4 * (T % 8) + e)) & 1).bool()
    # (compared where the forward read-out can see the unit at all: gelu(u) of a clearly negative u is below the read-out's threshold)
    live_all = ((a.float() @ w1.float().t() + b1).abs() > 0.05) & live_fwd
    assert torch.equal(saved[live_all], fwd_mask[live_all])
    assert abs(1.0 - saved.float().mean().item() - p) < 0.01
    # backward: a one-hot gradient row by row
    w2 = rnd(C, H, scale=0.5, seed=61, dtype=BF).abs() + 0.1
    w2 = w2.to(BF)
    bwd_kept = torch.zeros(M, H, dtype=torch.bool, device=DEV)
    u = a.float() @ w1.float().t() + b1
    for row in range(0, M, 7):                                   # every 7th row: 19 launches
        gm = torch.zeros(M, C, dtype=BF, device=DEV)
        gm[row] = 1.0
        da = torch.empty(M, C, dtype=BF, device=DEV)
        dw1, db1 = torch.zeros(H, C, device=DEV), torch.zeros(H, device=DEV)
        dw2, db2 = torch.zeros(C, H, device=DEV), torch.zeros(C, device=DEV)
        ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2, mask_bits=bits)
        bwd_kept[row] = db1 != 0
        sure = (u[row].abs() > 0.05) & (u[row] > -3)             # gelu'(u) clearly non-zero there
        assert torch.equal(bwd_kept[row][sure], fwd_mask[row][sure]), row
    with pytest.raises(Exception, match="mask_bits"):          # hidden dropout on and no saved words: an error, never a silent default
        ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2)
    # a different seed word -> a different mask
    rng2 = ops.new_rng_state(78, DEV)
    d2 = ops.mlp_desc(cc, M, C, H, ops.drop_desc(rng2, 21, p, 25, 0.0, 16), ops.drop_desc(rng2, 22, 0.0, 26, 0.0, 16))
    w2s = torch.zeros(C, H, device=DEV)
    w2s[torch.arange(C), torch.arange(C)] = 1.0
    ya, yb = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    ops.mlp_fwd(d, a, r, w1, b1, w2s.to(BF), b2, ya)
    ops.mlp_fwd(d2, a, r, w1, b1, w2s.to(BF), b2, yb)
    assert ((ya != 0) != (yb != 0)).float().mean().item() > 0.2


@pytest.mark.parametrize("M,p_drop", [(4096, 0.0), (9216 + 48, 0.2)])
def test_fused_mlp_backward_with_norm2_backward_folded_in(ops, M, p_drop):
    """focal_mlp_bwd with ln_x: the backward of the LayerNorm in front of the branch (norm2, models/SwinModules.py:339-341) on the kernel's
    fp32 dL/da2 -- g += dLN, gm_next = bf16(g x mask of the attention branch), dgamma / dbeta -- must equal the two-launch form
    (focal_mlp_bwd -> dL/da2 in bf16 -> focal_layernorm_bwd) to the rounding of that bf16 intermediate; the parameter gradients of the
    MLP itself must be identical (same code path).  Ragged M (not a multiple of the 128-token tile) and dropout masks included."""
    C, H, L = 64, 256, 576
    a, w1, b1, w2, b2, r = _operands(M, seed=70)
    gm = rnd(M, C, scale=0.5, seed=71, dtype=BF)
    x_mid = rnd(M, C, seed=72)
    gamma, beta = rnd(C, seed=73).abs() + 0.5, rnd(C, seed=74)
    cc = ops.code(BF)
    rng = ops.new_rng_state(99, DEV)
    drop_h = ops.drop_desc(rng, 31, p_drop, 35, 0.0, L)
    next_mask = ops.drop_desc(rng, 33, p_drop, 37, 0.1 if p_drop else 0.0, L)
    d = ops.mlp_desc(cc, M, C, H, drop_h, ops.drop_desc(rng, 32, 0.0, 36, 0.0, L))
    bits = ops.mlp_mask_bits(d, DEV)
    a2, stats = ops.layernorm_fwd(x_mid, gamma, beta, BF)
    y = torch.empty(M, C, device=DEV)
    ops.mlp_fwd(d, a2, r, w1, b1, w2, b2, y, mask_bits=bits)
    g0 = rnd(M, C, seed=75)

    def grads():
        return (torch.zeros(H, C, device=DEV), torch.zeros(H, device=DEV), torch.zeros(C, H, device=DEV), torch.zeros(C, device=DEV),
                torch.zeros(C, device=DEV), torch.zeros(C, device=DEV))
    # two launches
    dw1, db1, dw2, db2, dg, dbt = grads()
    da = torch.empty(M, C, dtype=BF, device=DEV)
    ops.mlp_bwd(d, gm, a2, w1, b1, w2, da, dw1, db1, dw2, db2, mask_bits=bits)
    g_ref, gmn_ref = g0.clone(), torch.empty(M, C, dtype=BF, device=DEV)
    ops.layernorm_bwd(da, x_mid, stats, gamma, g_ref, True, dg, dbt, dx_masked=gmn_ref, mask=next_mask)
    # one launch, gm_next written over gm (what the engine does)
    ew1, eb1, ew2, eb2, eg, ebt = grads()
    g_one, gm_io = g0.clone(), gm.clone()
    ops.mlp_bwd(d, gm_io, a2, w1, b1, w2, None, ew1, eb1, ew2, eb2, mask_bits=bits,
                ln=dict(x=x_mid, stats=stats, gamma=gamma, g=g_one, gm_next=gm_io, next_mask=next_mask, dgamma=eg, dbeta=ebt))
    assert rel_err(ew1, dw1) < 1e-5 and rel_err(ew2, dw2) < 1e-5 and rel_err(eb1, db1) < 1e-5 and rel_err(eb2, db2) < 1e-5
    # round 5: the weight gradients through the workspace + reduce launch instead of 33 MB of atomics (what the engine does), accumulating
    # into buffers that already hold a gradient: the same sums
    pw1, pb1, pw2, pb2, pg_, pbt = grads()
    pw1.fill_(0.25), pw2.fill_(-0.5)
    ws = ops.mlp_bwd_partials(d, DEV)
    assert ws.numel() == min((M + 127) // 128, 256) * 2 * C * H
    g_p, gm_p = g0.clone(), gm.clone()
    ops.mlp_bwd(d, gm_p, a2, w1, b1, w2, None, pw1, pb1, pw2, pb2, mask_bits=bits, partials=ws,
                ln=dict(x=x_mid, stats=stats, gamma=gamma, g=g_p, gm_next=gm_p, next_mask=next_mask, dgamma=pg_, dbeta=pbt))
    assert rel_err(pw1 - 0.25, ew1) < 1e-5 and rel_err(pw2 + 0.5, ew2) < 1e-5 and rel_err(pb1, eb1) < 1e-5 and rel_err(pb2, eb2) < 1e-5
    assert torch.equal(g_p, g_one) and torch.equal(gm_p, gm_io)
    assert rel_err(g_one - g0, g_ref - g0) < 6e-3            # (the two-launch form rounds dL/da2 to bf16 on the way)
    assert rel_err(eg, dg) < 4e-3 and rel_err(ebt, dbt) < 4e-3
    same_mask = ((gm_io.float() == 0) == (gmn_ref.float() == 0)) | (g_ref.abs() < 1e-3)
    assert same_mask.float().mean().item() > 0.999           # the attention branch's dropout x drop-path mask is the same function
    assert rel_err(gm_io.float(), gmn_ref.float()) < 1e-2
    # and against torch autograd through LayerNorm + MLP in fp32 (dropout off only)
    if p_drop == 0.0:
        xt = x_mid.clone().requires_grad_(True)
        gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        a_t = F.layer_norm(xt, (C,), gt, bt, 1e-5)
        yt = F.gelu(a_t.to(BF).float() @ w1.float().t() + b1) @ w2.float().t() + b2
        yt.backward(gm.float())
        assert rel_err(g_one - g0, xt.grad) < 1.5e-2 and rel_err(eg, gt.grad) < 1.5e-2 and rel_err(ebt, bt.grad) < 1e-2


@pytest.mark.parametrize("M,drop,ln", [(4608, False, True), (36864 + 48, False, False), (1000, True, True), (147456, True, True), (16, False, True)])
def test_proj_and_norm2_in_front_of_the_fused_mlp_equal_the_two_launches(ops, M, drop, ln, monkeypatch):
    """focal_mlp_proj_fwd (round 6): x_mid = x + drop(o Wp^T + bp), a2 = norm2(x_mid) and the MLP branch in ONE launch against
    focal_linear_resid_ln_fwd + focal_mlp_fwd.  Masks off: every output bit-identical (the in-kernel LayerNorm repeats the GEMM epilogue's
    summation tree and roundings).  Masks on: x_mid to an ulp (hipcc contracts the residual expression of the GEMM epilogue for three of a
    lane's four columns), everything behind it to that ulp's consequences."""
    from focal_amd._lib import ACT_NONE, EPI_RESIDUAL
    C = 64
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    monkeypatch.setenv("FOCAL_MLP_PROJ", "0")
    assert not ops.mlp_proj_supported(BF, C, 4 * C)
    monkeypatch.delenv("FOCAL_MLP_PROJ")
    assert ops.mlp_proj_supported(BF, C, 4 * C) and not ops.mlp_proj_supported(BF, 128, 512)
    _, w1, b1, w2, b2, _ = _operands(M, seed=50)
    o, x = rnd(M, C, seed=61, dtype=BF), rnd(M, C, seed=62)
    wp, bp = rnd(C, C, scale=C ** -0.5, seed=63, dtype=BF), rnd(C, scale=0.3, seed=64)
    g2, bt2 = rnd(C, seed=65) * 0.2 + 1.0, rnd(C, seed=66) * 0.1
    gn, btn = rnd(C, seed=67) * 0.2 + 1.0, rnd(C, seed=68) * 0.1
    rng = ops.new_rng_state(4711 + M, DEV)
    dp = ops.drop_desc(rng, 5, 0.2, 9, 0.1, 576) if drop else None
    dh = ops.drop_desc(rng, 6, 0.2, 10, 0.0, 1) if drop else None
    do = ops.drop_desc(rng, 7, 0.2, 11, 0.1, 576) if drop else None
    d_proj = ops.linear_desc(cc, M, C, C, cc, f32, ACT_NONE, EPI_RESIDUAL, out_drop=dp)
    d = ops.mlp_desc(cc, M, C, 4 * C, dh, do)
    # the two launches
    xm0 = torch.empty(M, C, device=DEV)
    a20, st20 = ops.linear_resid_ln_fwd(d_proj, o, wp, bp, x, xm0, g2, bt2, BF)
    y0 = torch.empty(M, C, device=DEV)
    bits0 = ops.mlp_mask_bits(d, DEV)
    n0 = ops.mlp_fwd(d, a20, xm0, w1, b1, w2, b2, y0, next_ln=(gn, btn) if ln else None, mask_bits=bits0)
    # one launch
    xm1 = torch.full((M, C), 5.0, device=DEV)
    y1 = torch.full((M, C), 5.0, device=DEV)
    bits1 = ops.mlp_mask_bits(d, DEV)
    (a21, st21), n1 = ops.mlp_proj_fwd(d, o, x, wp, bp, dp, g2, bt2, xm1, w1, b1, w2, b2, y1, next_ln=(gn, btn) if ln else None, mask_bits=bits1)
    torch.cuda.synchronize()
    if not drop:
        assert torch.equal(xm1, xm0) and torch.equal(st21, st20) and torch.equal(a21, a20)
        assert torch.equal(y1, y0)
        if ln:
            assert torch.equal(n1[0], n0[0]) and torch.equal(n1[1], n0[1])
    else:
        assert (xm1 - xm0).abs().max().item() <= 2.5e-7 * xm0.abs().max().item()
        assert (a21 != a20).float().mean().item() < 2e-3 and torch.allclose(st21, st20, rtol=1e-5, atol=1e-6)
        assert torch.equal(bits1, bits0)
        assert rel_err(y1, y0) < 2e-3
        if ln:
            assert rel_err(n1[0].float(), n0[0].float()) < 4e-3
    ref = x + o.float() @ wp.float().t() + bp
    if not drop:
        assert rel_err(xm1, ref) < 2e-3
