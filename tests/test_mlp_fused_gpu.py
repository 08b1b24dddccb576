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


def test_fused_mlp_with_dropout_equals_unfused_hip_path(ops):
    """Same masks (hash of seed / stream / element index), same roundings: the fused kernels must reproduce the unfused chain --
    forward output and LayerNorm, and the backward's da / dW / db -- to accumulation-order noise."""
    from focal_amd._lib import ACT_GELU, EPI_GELU, EPI_RESIDUAL
    M, C, L = 9216, 64, 576
    a, w1, b1, w2, b2, r = _operands(M, seed=40)
    rng = ops.new_rng_state(1234, DEV)
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    drop_h = ops.drop_desc(rng, 11, 0.2, 15, 0.0, L)
    drop_o = ops.drop_desc(rng, 12, 0.2, 16, 0.1, L)
    # ---- unfused chain
    d1 = ops.linear_desc(cc, M, 4 * C, C, cc, cc, 0, EPI_GELU, out_drop=drop_h)
    h, hg = torch.empty(M, 4 * C, dtype=BF, device=DEV), torch.empty(M, 4 * C, dtype=BF, device=DEV)
    ops.linear_fwd(d1, a, w1, b1, None, h, hg)
    d2 = ops.linear_desc(cc, M, C, 4 * C, cc, f32, ACT_GELU, EPI_RESIDUAL, out_drop=drop_o)
    y_ref = torch.empty(M, C, device=DEV)
    ops.linear_fwd(d2, h, w2, b2, r, y_ref)
    assert (h == 0).float().mean().item() > 0.15  # dropout really on
    # ---- fused forward
    d = ops.mlp_desc(cc, M, C, 4 * C, drop_h, drop_o)
    y = torch.empty(M, C, device=DEV)
    ops.mlp_fwd(d, a, r, w1, b1, w2, b2, y)
    assert rel_err(y, y_ref) < 2e-5, rel_err(y, y_ref)
    # ---- backward: gm = bf16(g x out mask), as the LayerNorm backward / mask_cast hands it over
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
    assert rel_err(dw2, dw2_ref) < 2e-4 and rel_err(db2, db2_ref) < 2e-5
    assert rel_err(dw1, dw1_ref) < 6e-3 and rel_err(db1, db1_ref) < 6e-3   # the unfused chain multiplies by the bf16-ROUNDED saved derivative, the fused one by the fp32 value
