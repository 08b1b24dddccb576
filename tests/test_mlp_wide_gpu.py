"""The Swin MLP branch at 128 / 256 channels as one launch (focal_mlp_wide_fwd, round 6; reference: models/SwinModules.py:18-34 +
:339-341) through the C ABI: bit-identical to the two launches it replaces (focal_linear_fwd with the GELU epilogue, then
focal_linear_fwd / focal_linear_resid_ln_fwd with the residual epilogue) -- every output, every dropout / drop-path mask on and off,
ragged row counts -- and against torch fp32 of the same expression on the bf16-rounded operands."""
import os

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


def _operands(M, C, seed=0):
    a = rnd(M, C, seed=seed + 1, dtype=BF)
    w1, b1 = rnd(4 * C, C, scale=C ** -0.5, seed=seed + 2, dtype=BF), rnd(4 * C, scale=0.3, seed=seed + 3)
    w2, b2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=seed + 4, dtype=BF), rnd(C, scale=0.3, seed=seed + 5)
    r = rnd(M, C, seed=seed + 6)
    return a, w1, b1, w2, b2, r


@pytest.fixture(scope="module")
def ops():
    from focal_amd import ops as o
    return o


def _two_launches(ops, M, C, a, w1, b1, w2, b2, r, drop_h, drop_o, ln):
    """(h, hg, y, y_ln, stats) of the launches the wide kernel replaces: fc1 with the GELU epilogue, fc2 with the residual epilogue and -- as
    the Swin engine runs them -- the next LayerNorm in fc2's epilogue at 128 channels, as a stand-alone launch at 256."""
    from focal_amd._lib import ACT_GELU, ACT_NONE, EPI_GELU, EPI_RESIDUAL
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    d1 = ops.linear_desc(cc, M, 4 * C, C, cc, cc, ACT_NONE, EPI_GELU, out_drop=drop_h)
    d2 = ops.linear_desc(cc, M, C, 4 * C, cc, f32, ACT_GELU, EPI_RESIDUAL, out_drop=drop_o)
    h, hg = torch.empty(M, 4 * C, dtype=BF, device=DEV), torch.empty(M, 4 * C, dtype=BF, device=DEV)
    ops.linear_fwd(d1, a, w1, b1, None, h, hg)
    y = torch.empty(M, C, device=DEV)
    if ln is not None and C == 128:
        y_ln, stats = ops.linear_resid_ln_fwd(d2, h, w2, b2, r, y, ln[0], ln[1], BF)
        return h, hg, y, y_ln, stats
    ops.linear_fwd(d2, h, w2, b2, r, y)
    if ln is not None:
        y_ln, stats = ops.layernorm_fwd(y, ln[0], ln[1], BF)
        return h, hg, y, y_ln, stats
    return h, hg, y, None, None


@pytest.mark.parametrize("C,M,drop,ln", [(128, 36864, True, True), (128, 36864, False, False), (128, 1000, True, True), (128, 96, False, True),
                                         (256, 9216, True, False), (256, 18432, False, False), (256, 1000, True, False), (256, 50, False, False),
                                         (256, 9216, False, True), (256, 18432, True, True), (256, 1000, False, True)])
def test_one_launch_equals_the_two_launches_bit_for_bit(ops, C, M, drop, ln, monkeypatch):
    monkeypatch.setenv("FOCAL_MLP_WIDE", "0")
    assert not ops.mlp_wide_supported(BF, C, 4 * C)   # (the same-box A/B switch)
    monkeypatch.setenv("FOCAL_MLP_WIDE", str(C))
    assert ops.mlp_wide_supported(BF, C, 4 * C) and not ops.mlp_wide_supported(BF, 384 - C, 4 * (384 - C))
    monkeypatch.delenv("FOCAL_MLP_WIDE")
    assert ops.mlp_wide_supported(BF, C, 4 * C) and not ops.mlp_wide_supported(torch.float32, C, 4 * C) and not ops.mlp_wide_supported(BF, 64, 256)
    a, w1, b1, w2, b2, r = _operands(M, C, seed=3 * C + M)
    rng = ops.new_rng_state(20260 + M, DEV)
    drop_h = ops.drop_desc(rng, 17, 0.2, 21, 0.0, 1) if drop else None
    drop_o = ops.drop_desc(rng, 18, 0.2, 22, 0.1, 9) if drop else None
    lnp = (rnd(C, seed=11) * 0.2 + 1.0, rnd(C, seed=12) * 0.1) if ln else None
    h0, hg0, y0, yln0, st0 = _two_launches(ops, M, C, a, w1, b1, w2, b2, r, drop_h, drop_o, lnp)
    d = ops.mlp_desc(ops.code(BF), M, C, 4 * C, drop_h, drop_o)
    h, hg = torch.full((M, 4 * C), 7.0, dtype=BF, device=DEV), torch.full((M, 4 * C), 7.0, dtype=BF, device=DEV)
    y = torch.full((M, C), 7.0, device=DEV)
    out = ops.mlp_wide_fwd(d, a, r, w1, b1, w2, b2, y, h, hg, next_ln=lnp)
    torch.cuda.synchronize()
    assert torch.equal(h, h0) and torch.equal(hg, hg0)
    if not drop:
        assert torch.equal(y, y0)
        if ln:
            assert torch.equal(out[0], yln0) and torch.equal(out[1], st0)
    else:
        # masks on: y = resid + v * row mask * element mask is one rounding or two depending on whether hipcc contracts it to an FMA, and it
        # decides that per kernel instantiation (the ring and the one-tile GEMM already differ from each other in one of a lane's four
        # columns): equal to an ulp of y, and the LayerNorm outputs to a bf16 ulp on a handful of elements
        assert (y - y0).abs().max().item() <= 2.5e-7 * y0.abs().max().item()
        assert (y != y0).float().mean().item() < 0.08
        if ln:
            assert (out[0].float() - yln0.float()).abs().max().item() <= 2.0 ** -7 * yln0.float().abs().max().item()
            assert (out[0] != yln0).float().mean().item() < 1e-3 and torch.allclose(out[1], st0, rtol=1e-5, atol=1e-6)
    if not ln:
        assert out is None
    if drop:  # the masks are on: a fifth of the hidden units are zero in h and in hg alike
        z = (hg == 0).float().mean().item()
        assert 0.18 < z < 0.22, z
        assert bool(((hg == 0) <= (h == 0)).all())
    if not drop:
        hh = F.gelu(a.float() @ w1.float().t() + b1)
        assert rel_err(h.float(), hh) < 4e-3
        ref = r + hh.to(BF).float() @ w2.float().t() + b2
        assert rel_err(y, ref) < 3e-3


def test_wide_mlp_rejects_what_it_was_not_built_for(ops):
    from focal_amd._lib import FocalHipError
    a, w1, b1, w2, b2, r = _operands(64, 256)
    d = ops.mlp_desc(ops.code(BF), 64, 256, 1024)
    y, h, hg = torch.empty(64, 256, device=DEV), torch.empty(64, 1024, dtype=BF, device=DEV), torch.empty(64, 1024, dtype=BF, device=DEV)
    d64 = ops.mlp_desc(ops.code(BF), 64, 64, 256)
    with pytest.raises((FocalHipError, AssertionError)):
        ops.mlp_wide_fwd(d64, a[:, :64].contiguous(), r[:, :64].contiguous(), w1, b1, w2, b2, y[:, :64].contiguous(), h, hg)


def test_encoder_with_the_one_launch_mlp_equals_the_default(cfg, monkeypatch):
    """The one-launch MLP (the default) against FOCAL_MLP_WIDE=0 through the Swin engine (stages 1-2 of both modality encoders, dropout off): embeddings equal to the default two-launch
    form up to the order of mod_in's split-K atomics; h and hg are the same tensors, so the backward pass cannot tell (gradients equal up to the order of the weight
    gradients' fp32 atomics, which differs between any two runs)."""
    from test_swt_parity_gpu import build, inputs

    def run():
        args, net, focal, loss_fn = build(cfg, "bf16")
        net.train()
        x1, x2 = inputs(cfg)
        f1, f2 = focal(x1, x2, proj_head=True)
        loss = loss_fn(f1, f2)
        net.arena().zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return {m: f1[m].detach().clone() for m in f1}, net.arena().grad.clone()
    monkeypatch.setenv("FOCAL_MLP_WIDE", "0")
    e0, g0 = run()
    monkeypatch.delenv("FOCAL_MLP_WIDE")
    e1, g1 = run()
    from conftest import record_observed
    record_observed("mlp_wide.encoder.emb_diff_over_max", max(((e0[m] - e1[m]).abs().max() / e0[m].abs().max()).item() for m in e0))
    record_observed("mlp_wide.encoder.grad_diff_over_max", ((g0 - g1).abs().max() / g0.abs().max()).item())
    for m in e0:  # (mod_in's split-K product sums its slices with fp32 atomics: equal to their order)
        assert (e0[m] - e1[m]).abs().max().item() <= 1e-5 * e0[m].abs().max().item()
    # (mod_in's split-K atomics move dL/dfeat in the last fp32 bit between ANY two runs; the bf16 casts of the backward operands turn a
    # few of those into 2^-8 steps: the gradients of two runs of the SAME form differ by this much)
    assert (g0 - g1).abs().max().item() <= 1e-2 * g0.abs().max().item()   # (observed 0.9e-3 .. 2.6e-3 over six runs)


@pytest.mark.parametrize("C,M,ln,drop", [(128, 36864, True, True), (128, 1000, True, False), (128, 96, False, False), (256, 9216, False, True), (256, 1000, False, False)])
def test_backward_data_path_in_one_launch_equals_the_two_launches(ops, C, M, ln, drop):
    """focal_mlp_wide_bwd_data against focal_linear_bwd_data (GELU derivative) + focal_linear_bwd_data / focal_linear_bwd_data_ln: du, dc or
    the LayerNorm backward's g / g_masked / dgamma / dbeta.  The products are the same sums in the same 32-term groups; the first one's k slots
    inside an MFMA are permuted, so the comparison is to fp32 accumulation order (1e-6-class) on top of bf16 outputs' last-bit flips."""
    from focal_amd._lib import ACT_GELU, ACT_NONE
    cc = ops.code(BF)
    H = 4 * C
    gm = rnd(M, C, scale=0.5, seed=41, dtype=BF)
    hg = rnd(M, H, scale=0.7, seed=42, dtype=BF)
    w1, w2 = rnd(H, C, scale=C ** -0.5, seed=43, dtype=BF), rnd(C, H, scale=H ** -0.5, seed=44, dtype=BF)
    d2 = ops.linear_desc(cc, M, C, H, cc, cc, ACT_GELU)
    d1 = ops.linear_desc(cc, M, H, C, cc, cc, ACT_NONE)
    du0 = torch.empty(M, H, dtype=BF, device=DEV)
    ops.linear_bwd_data(d2, gm, w2, hg, du0)
    dw = ops.mlp_desc(cc, M, C, H)
    du = torch.full((M, H), 3.0, dtype=BF, device=DEV)
    if not ln:
        dc0 = torch.empty(M, C, dtype=BF, device=DEV)
        ops.linear_bwd_data(d1, du0, w1, None, dc0)
        dc = torch.full((M, C), 3.0, dtype=BF, device=DEV)
        ops.mlp_wide_bwd_data(dw, gm, hg, w1, w2, du, dc=dc)
        torch.cuda.synchronize()
    else:
        x = rnd(M, C, seed=45)
        stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
        gamma = rnd(C, seed=46) * 0.2 + 1.0
        rng = ops.new_rng_state(99, DEV)
        mask = ops.drop_desc(rng, 33, 0.2, 37, 0.1, 9) if drop else None
        g0, g1 = rnd(M, C, seed=47), rnd(M, C, seed=47)
        gmk0, gmk1 = torch.empty(M, C, dtype=BF, device=DEV), torch.empty(M, C, dtype=BF, device=DEV)
        dg0, db0, dg1, db1 = (torch.zeros(C, device=DEV) for _ in range(4))
        ops.linear_bwd_data_ln(d1, du0, w1, x, stats, gamma, g0, dg0, db0, g_masked=gmk0, mask=mask)
        ops.mlp_wide_bwd_data(dw, gm, hg, w1, w2, du, ln=dict(x=x, stats=stats, gamma=gamma, g=g1, g_masked=gmk1, mask=mask, dgamma=dg1, dbeta=db1))
        torch.cuda.synchronize()
    # du: bf16 of the same fp32 product up to accumulation order: a last-bit flip here and there
    assert (du != du0).float().mean().item() < 0.02
    assert (du.float() - du0.float()).abs().max().item() <= 2.0 ** -7 * du0.float().abs().max().item()
    if not ln:
        assert rel_err(dc.float(), dc0.float()) < 3e-3 and (dc.float() - dc0.float()).abs().max().item() <= 2.0 ** -6 * dc0.float().abs().max().item()
        ref = (gm.float() @ w2.float()) * hg.float()
        assert rel_err(du.float(), ref) < 4e-3
        assert rel_err(dc.float(), ref.to(BF).float() @ w1.float()) < 6e-3
    else:
        assert rel_err(g1, g0) < 1e-3 and rel_err(dg1, dg0) < 2e-3 and rel_err(db1, db0) < 2e-3
        assert (gmk1 != gmk0).float().mean().item() < 0.02
        if drop:
            assert bool(((gmk0 == 0) == (gmk1 == 0)).float().mean().item() > 0.999)


def test_encoder_with_the_one_launch_backward_equals_the_default(cfg, monkeypatch):
    """FOCAL_MLP_WIDE_BWD=1 through the Swin engine (the one-launch backward data path of stages 1-2; off by default: it does not pay inside
    the step): same embeddings, gradients equal to run-to-run noise of the bf16 backward."""
    from test_swt_parity_gpu import build, inputs

    def run():
        args, net, focal, loss_fn = build(cfg, "bf16")
        net.train()
        x1, x2 = inputs(cfg)
        f1, f2 = focal(x1, x2, proj_head=True)
        loss = loss_fn(f1, f2)
        net.arena().zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return {m: f1[m].detach().clone() for m in f1}, net.arena().grad.clone()
    from focal_amd import ops as o
    assert not o.mlp_wide_bwd_supported(BF, 128, 512)
    e0, g0 = run()
    monkeypatch.setenv("FOCAL_MLP_WIDE_BWD", "1")
    assert o.mlp_wide_bwd_supported(BF, 128, 512) and o.mlp_wide_bwd_supported(BF, 256, 1024)
    e1, g1 = run()
    for m in e0:
        assert (e0[m] - e1[m]).abs().max().item() <= 1e-5 * e0[m].abs().max().item()
    assert (g0 - g1).abs().max().item() <= 1e-2 * g0.abs().max().item()
    assert ((g0 - g1).norm() / g0.norm()).item() < 5e-3


@pytest.mark.parametrize("C,M,drop,ln", [(128, 36864, False, True), (128, 1000, False, False), (128, 9216, True, True), (256, 9216, False, True), (256, 18432, False, False),
                                         (256, 1000, True, True), (256, 50, False, True)])
def test_proj_and_norm2_in_front_of_the_wide_mlp_equal_their_own_launches(ops, C, M, drop, ln, monkeypatch):
    """focal_mlp_wide_proj_fwd: x_mid = x + drop(o Wp^T + bp), a2 = norm2(x_mid) and the MLP branch in ONE launch against
    focal_linear_resid_ln_fwd (128 channels) / focal_linear_fwd + focal_layernorm_fwd (256) followed by focal_mlp_wide_fwd.  Masks off: every
    output bit-identical (the in-kernel LayerNorm repeats the replaced launch's summation tree and roundings)."""
    from focal_amd._lib import ACT_NONE, EPI_RESIDUAL
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    monkeypatch.setenv("FOCAL_MLP_PROJ", "0")
    assert not ops.mlp_wide_proj_supported(BF, C, 4 * C)
    monkeypatch.delenv("FOCAL_MLP_PROJ")
    assert ops.mlp_wide_proj_supported(BF, C, 4 * C)
    _, w1, b1, w2, b2, _ = _operands(M, C, seed=70 + C)
    o, x = rnd(M, C, seed=81, dtype=BF), rnd(M, C, seed=82)
    wp, bp = rnd(C, C, scale=C ** -0.5, seed=83, dtype=BF), rnd(C, scale=0.3, seed=84)
    g2, bt2 = rnd(C, seed=85) * 0.2 + 1.0, rnd(C, seed=86) * 0.1
    lnp = (rnd(C, seed=87) * 0.2 + 1.0, rnd(C, seed=88) * 0.1) if ln else None
    rng = ops.new_rng_state(815 + M, DEV)
    dp = ops.drop_desc(rng, 5, 0.2, 9, 0.1, 64) if drop else None
    dh = ops.drop_desc(rng, 6, 0.2, 10, 0.0, 1) if drop else None
    do = ops.drop_desc(rng, 7, 0.2, 11, 0.1, 64) if drop else None
    d_proj = ops.linear_desc(cc, M, C, C, cc, f32, ACT_NONE, EPI_RESIDUAL, out_drop=dp)
    d = ops.mlp_desc(cc, M, C, 4 * C, dh, do)
    xm0 = torch.empty(M, C, device=DEV)
    if C == 128:
        a20, st20 = ops.linear_resid_ln_fwd(d_proj, o, wp, bp, x, xm0, g2, bt2, BF)
    else:
        ops.linear_fwd(d_proj, o, wp, bp, x, xm0)
        a20, st20 = ops.layernorm_fwd(xm0, g2, bt2, BF)
    h0, hg0 = torch.empty(M, 4 * C, dtype=BF, device=DEV), torch.empty(M, 4 * C, dtype=BF, device=DEV)
    y0 = torch.empty(M, C, device=DEV)
    n0 = ops.mlp_wide_fwd(d, a20, xm0, w1, b1, w2, b2, y0, h0, hg0, next_ln=lnp)
    xm1, y1 = torch.full((M, C), 5.0, device=DEV), torch.full((M, C), 5.0, device=DEV)
    h1, hg1 = torch.full((M, 4 * C), 5.0, dtype=BF, device=DEV), torch.full((M, 4 * C), 5.0, dtype=BF, device=DEV)
    (a21, st21), n1 = ops.mlp_wide_proj_fwd(d, o, x, wp, bp, dp, g2, bt2, xm1, w1, b1, w2, b2, y1, h1, hg1, next_ln=lnp)
    torch.cuda.synchronize()
    if not drop:
        assert torch.equal(xm1, xm0)
        assert torch.equal(st21, st20) and torch.equal(a21, a20)
        assert torch.equal(h1, h0) and torch.equal(hg1, hg0) and torch.equal(y1, y0)
        if ln:
            assert torch.equal(n1[0], n0[0]) and torch.equal(n1[1], n0[1])
        assert rel_err(xm1, x + o.float() @ wp.float().t() + bp) < 2e-3
    else:
        assert (xm1 - xm0).abs().max().item() <= 2.5e-7 * xm0.abs().max().item()
        assert (a21 != a20).float().mean().item() < 2e-3 and torch.allclose(st21, st20, rtol=1e-5, atol=1e-6)
        assert rel_err(y1, y0) < 2e-3 and rel_err(h1.float(), h0.float()) < 2e-3
