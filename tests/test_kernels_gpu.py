"""Per-kernel parity of the HIP ops (through the C ABI) against plain PyTorch references of the same op.
fp32 mode must agree to ~1e-4 relative (exact-fp32 MFMA); bf16 mode to bf16 operand rounding."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 2e-4, torch.bfloat16: 2.5e-2}


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from focal_amd import ops as o
    return o


# ---------------------------------------------------------------------------------------------- linear family
@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4608, 192, 64), (1000, 64, 256), (144, 768, 256), (4608, 128, 512), (36, 64, 64)])
def test_linear_fwd_plain(ops, ct, M, N, K):
    x, w, b = rnd(M, K, seed=1, dtype=ct), rnd(N, K, scale=K ** -0.5, seed=2, dtype=ct), rnd(N, seed=3)
    y, _ = ops.linear(x, w, b, compute=ct)
    ref = x.float() @ w.float().t() + b
    assert rel_err(y.float(), ref) < (1e-5 if ct == torch.float32 else 6e-3)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_linear_gelu_epilogue_and_residual(ops, ct):
    """Swin MLP forward: (h, h') = gelu(a W1^T + b1) and its derivative from ONE kernel; y = r + h W2^T + b2."""
    from focal_amd._lib import ACT_GELU, EPI_GELU, EPI_RESIDUAL
    M, C = 1152, 64
    a, w1, b1 = rnd(M, C, seed=4, dtype=ct), rnd(4 * C, C, scale=C ** -0.5, seed=5, dtype=ct), rnd(4 * C, seed=6)
    hg = torch.empty(M, 4 * C, dtype=ct, device=DEV)
    h, _ = ops.linear(a, w1, b1, compute=ct, epilogue=EPI_GELU, act_grad=hg)
    u = (a.float() @ w1.float().t() + b1).requires_grad_(True)
    href = F.gelu(u)
    href.sum().backward()
    tol = 2e-6 if ct == torch.float32 else 5e-3
    assert rel_err(h.float(), href) < tol and rel_err(hg.float(), u.grad) < tol
    if ct == torch.float32:  # the erf approximation itself: absolute error << 1e-6
        assert (h - href).abs().max().item() < 2e-6 and (hg - u.grad).abs().max().item() < 2e-6
    w2, b2, r = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=7, dtype=ct), rnd(C, seed=8), rnd(M, C, seed=9)
    y, _ = ops.linear(h, w2, b2, compute=ct, y_dtype=torch.float32, resid=r, act_in=ACT_GELU, epilogue=EPI_RESIDUAL)
    ref = r + h.float() @ w2.float().t() + b2
    assert rel_err(y, ref) < (1e-5 if ct == torch.float32 else 4e-3)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_linear_split_k_and_relu(ops, ct):
    from focal_amd._lib import EPI_RELU
    B, N, K = 16, 256, 4608
    x, w, b = rnd(B, K, seed=8), rnd(N, K, scale=K ** -0.5, seed=9, dtype=ct), rnd(N, seed=10)
    y, _ = ops.linear(x, w, b, compute=ct, y_dtype=torch.float32, splits=9)
    xr = x if ct == torch.float32 else x.bfloat16().float()
    ref = xr @ w.float().t() + b
    assert rel_err(y, ref) < (1e-5 if ct == torch.float32 else 4e-3)
    y2, _ = ops.linear(x[:, :256].contiguous(), w[:, :256].contiguous(), b, compute=ct, y_dtype=torch.float32, epilogue=EPI_RELU)
    ref2 = F.relu(xr[:, :256] @ w[:, :256].float().t() + b)
    assert rel_err(y2, ref2) < (1e-5 if ct == torch.float32 else 4e-3)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4608, 192, 64), (1000, 64, 256), (144, 256, 1024), (2304, 512, 128),
                                   (2304, 1024, 256), (4672, 256, 1024), (64, 1024, 256), (192, 64, 64)])  # whole 64-tiles: the LDS-DMA ring kernel in bf16
def test_linear_bwd(ops, ct, M, N, K):
    x = rnd(M, K, seed=11, dtype=ct)
    w = rnd(N, K, scale=K ** -0.5, seed=12, dtype=ct)
    dy = rnd(M, N, seed=13, dtype=ct)
    c = ops.code(ct)
    d = ops.linear_desc(c, M, N, K, c, c)
    dx = torch.empty(M, K, dtype=ct, device=DEV)
    ops.linear_bwd_data(d, dy, w, None, dx)
    assert rel_err(dx.float(), dy.float() @ w.float()) < (1e-5 if ct == torch.float32 else 6e-3)
    import ctypes
    from focal_amd import _lib
    ring = ct == torch.bfloat16 and M % 64 == 0 and N % 64 == 0 and K % 64 == 0   # the LDS-DMA ring kernel takes whole 64-tiles in bf16
    assert _lib.load().focal_linear_bwd_weight_kernel(ctypes.byref(d)) == (2 if ring else 1)
    dw = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    ops.linear_bwd_weight(d, dy, x, dw, db)
    assert rel_err(dw, dy.float().t() @ x.float()) < (2e-5 if ct == torch.float32 else 1e-4)
    assert rel_err(db, dy.float().sum(0)) < 2e-5
    # accumulation semantics: a second call adds
    ops.linear_bwd_weight(d, dy, x, dw, db)
    assert rel_err(dw, 2 * (dy.float().t() @ x.float())) < (2e-5 if ct == torch.float32 else 1e-4)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (128, 64, 192), (192, 192, 64), (320, 64, 64), (8256, 64, 64), (4160, 384, 128), (1088, 768, 256),
                                   (960, 1024, 1024), (640, 2048, 1024), (70400, 64, 128), (33024, 128, 64), (2112, 256, 704)])
def test_weight_gradient_ring_kernel_shapes(ops, M, N, K):
    """The LDS-DMA ring kernel over its launch plans: one token slice per 4-wave workgroup (few rows), two slices per 8-wave workgroup with
    the LDS reduction, slices of unequal or zero length (row counts that do not divide), 3 / 12 / 48 / 256 / 512 output tiles (the cap of
    one workgroup per CU, and beyond it), with and without the bias gradient, accumulating into a non-zero buffer."""
    import ctypes
    from focal_amd import _lib
    ct = torch.bfloat16
    x, dy = rnd(M, K, seed=21, dtype=ct), rnd(M, N, seed=22, dtype=ct)
    c = ops.code(ct)
    d = ops.linear_desc(c, M, N, K, c, c)
    assert _lib.load().focal_linear_bwd_weight_kernel(ctypes.byref(d)) == 2
    ref = dy.float().t() @ x.float()
    dw = torch.full((N, K), 0.5, device=DEV)
    db = torch.full((N,), -1.0, device=DEV)
    ops.linear_bwd_weight(d, dy, x, dw, db)
    assert rel_err(dw - 0.5, ref) < 2e-4
    assert rel_err(db + 1.0, dy.float().sum(0)) < 1e-4
    dw2 = torch.zeros(N, K, device=DEV)
    ops.linear_bwd_weight(d, dy, x, dw2, None)
    assert rel_err(dw2, ref) < 2e-4


@pytest.mark.parametrize("M,N,K", [(2304, 128, 128), (2304, 128, 512), (1000, 128, 128), (576, 256, 256), (592, 256, 1024), (64, 256, 256), (9216, 256, 1024),
                                   (36864, 128, 512), (4096 + 24, 128, 128)])   # (the last two: the ring kernel, full and ragged)
def test_linear_resid_ln_wide(ops, M, N, K):
    """focal_linear_resid_ln_fwd at 128 / 256 columns (bf16): y = resid + x w^T + b from the LDS-DMA GEMM with row-complete wave tiles,
    LayerNorm(y) and its statistics from the same epilogue = focal_linear_fwd followed by focal_layernorm_fwd (ragged M included)."""
    from focal_amd._lib import ACT_NONE, EPI_RESIDUAL
    ct = torch.bfloat16
    c, f32 = ops.code(ct), ops.code(torch.float32)
    assert ops.resid_ln_supported(c, N, K)
    x, w, b = rnd(M, K, seed=41, dtype=ct), rnd(N, K, scale=K ** -0.5, seed=42, dtype=ct), rnd(N, seed=43)
    resid = rnd(M, N, seed=44)
    gamma, beta = rnd(N, seed=45) * 0.1 + 1.0, rnd(N, seed=46) * 0.1
    d = ops.linear_desc(c, M, N, K, c, f32, ACT_NONE, EPI_RESIDUAL)
    y = torch.empty(M, N, device=DEV)
    y_ln, stats = ops.linear_resid_ln_fwd(d, x, w, b, resid, y, gamma, beta, ct)
    y_ref = torch.empty(M, N, device=DEV)
    ops.linear_fwd(d, x, w, b, resid, y_ref)
    ln_ref, st_ref = ops.layernorm_fwd(y_ref, gamma, beta, ct)
    assert rel_err(y, y_ref) < 1e-6
    assert rel_err(stats, st_ref) < 1e-5
    assert rel_err(y_ln.float(), ln_ref.float()) < 3e-3  # bf16 outputs of values that differ in the last fp32 bits
    t = resid + x.float() @ w.float().t() + b
    assert rel_err(y, t) < 1e-5
    assert rel_err(y_ln.float(), F.layer_norm(t, (N,), gamma, beta, 1e-5)) < 4e-3


@pytest.mark.parametrize("M,N,K,drop", [(2304, 192, 64, False), (2304, 384, 128, True), (1000, 512, 128, True), (64, 256, 64, False), (9216, 384, 128, False),
                                         (4672, 192, 64, True), (1152, 768, 256, True), (1000, 1024, 256, False), (64, 768, 256, True), (9216, 1024, 256, True),
                                         (73728, 192, 64, True), (4096 + 40, 192, 64, False)])   # (M >= 4096 at 64 channels: the ring kernel, full and ragged)
def test_linear_bwd_data_ln(ops, M, N, K, drop):
    """focal_linear_bwd_data_ln = focal_linear_bwd_data followed by focal_layernorm_bwd (accumulating into g, with the masked operand copy
    for the next branch): the LayerNorm backward as the epilogue of the dX GEMM (row-complete wave tiles), ragged M, with / without a mask."""
    ct = torch.bfloat16
    c = ops.code(ct)
    assert ops.bwd_data_ln_supported(c, N, K)
    dy, w = rnd(M, N, seed=51, dtype=ct), rnd(N, K, scale=N ** -0.5, seed=52, dtype=ct)
    x = rnd(M, K, seed=53) * 1.5 + 0.2
    gamma, beta = rnd(K, seed=54) * 0.1 + 1.0, rnd(K, seed=55) * 0.1
    g0 = rnd(M, K, seed=56)
    _, stats = ops.layernorm_fwd(x, gamma, beta, ct)
    rng = ops.new_rng_state(1234, DEV)
    mask = ops.drop_desc(rng, 5, 0.2, 9, 0.1, 576) if drop else None
    d = ops.linear_desc(c, M, N, K, c, c)
    # reference: the two kernels
    da = torch.empty(M, K, dtype=ct, device=DEV)
    ops.linear_bwd_data(d, dy, w, None, da)
    g_ref, gm_ref = g0.clone(), torch.empty(M, K, dtype=ct, device=DEV)
    dg_ref, db_ref = torch.full((K,), 0.5, device=DEV), torch.full((K,), -0.5, device=DEV)
    ops.layernorm_bwd(da, x, stats, gamma, g_ref, True, dg_ref, db_ref, dx_masked=gm_ref, mask=mask)
    # fused
    g, gm = g0.clone(), torch.empty(M, K, dtype=ct, device=DEV)
    dg, db = torch.full((K,), 0.5, device=DEV), torch.full((K,), -0.5, device=DEV)
    ops.linear_bwd_data_ln(d, dy, w, x, stats, gamma, g, dg, db, g_masked=gm, mask=mask)
    # (the two-kernel path rounds dy . w to bf16 in between, the fused one does not: agreement to that rounding)
    assert rel_err(g - g0, g_ref - g0) < 6e-3
    assert rel_err(gm.float(), gm_ref.float()) < 6e-3
    assert rel_err(dg - 0.5, dg_ref - 0.5) < 6e-3 and rel_err(db + 0.5, db_ref + 0.5) < 6e-3
    # and against fp32 torch: LayerNorm backward of da = dy w
    xt = x.clone().requires_grad_(True)
    gt = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    y = F.layer_norm(xt, (K,), gt, bt, 1e-5)
    y.backward(dy.float() @ w.float())
    assert rel_err(g - g0, xt.grad) < 2e-5 + 1e-3 * 0  # fp32 accumulate over bf16 operands: the products are exact in fp32
    assert rel_err(dg - 0.5, gt.grad) < 1e-4 and rel_err(db + 0.5, bt.grad) < 1e-4
    # no masked copy requested
    g2 = g0.clone()
    ops.linear_bwd_data_ln(d, dy, w, x, stats, gamma, g2, torch.zeros(K, device=DEV), torch.zeros(K, device=DEV))
    assert rel_err(g2, g) < 1e-6
    # g = None: the LayerNorm's input is a leaf nobody differentiates (norm1 of the first block behind the frozen patch embedding):
    # dgamma / dbeta exactly as before, no residual-stream gradient read or written
    dg3, db3 = torch.full((K,), 0.5, device=DEV), torch.full((K,), -0.5, device=DEV)
    ops.linear_bwd_data_ln(d, dy, w, x, stats, gamma, None, dg3, db3)
    assert rel_err(dg3 - 0.5, dg - 0.5) < 1e-5 and rel_err(db3 + 0.5, db + 0.5) < 1e-5
    with pytest.raises(Exception, match="g_masked without g"):
        ops.linear_bwd_data_ln(d, dy, w, x, stats, gamma, None, dg3, db3, g_masked=gm, mask=mask)


@pytest.mark.parametrize("rows,C,exclusive", [(2304, 128, True), (2304, 128, False), (64, 128, True), (1152, 256, True), (4672, 256, False),
                                              (36864, 128, True), (576, 256, True)])
def test_weight_gradient_group_kernel(ops, rows, C, exclusive):
    """focal_linear_bwd_weight_group: the four weight gradients of a Swin block (fc2 [C, 4C], fc1 [4C, C], proj [C, C], qkv [3C, C]) as
    one launch on 128 x 128 tiles = four focal_linear_bwd_weight calls: several token slices per tile (atomics) and one (plain
    read-add-write when `exclusive`), slices of unequal / zero length, accumulation into non-zero buffers, with and without bias."""
    ct = torch.bfloat16
    c = ops.code(ct)
    shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        assert ops.dw_group_supported(c, rows, N, K)
        dy, x = rnd(rows, N, seed=300 + i, dtype=ct), rnd(rows, K, seed=310 + i, dtype=ct)
        dw = torch.full((N, K), 0.25 * (i + 1), device=DEV)
        db = torch.full((N,), -1.0, device=DEV) if i != 2 else None
        items.append((dy, x, dw, db))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    wgs = ops.linear_bwd_weight_group_workgroups(c, items, exclusive)
    assert 0 < wgs <= max(256, sum((N // 128) * (K // 128) for N, K in shapes))
    ops.linear_bwd_weight_group(c, items, exclusive=exclusive)
    for i, ((dy, x, dw, db), (rw, rb)) in enumerate(zip(items, refs)):
        assert rel_err(dw - 0.25 * (i + 1), rw) < 2e-4, i
        if db is not None:
            assert rel_err(db + 1.0, rb) < 1e-4, i
    ops.linear_bwd_weight_group(c, items, exclusive=exclusive)  # accumulates
    for i, ((dy, x, dw, db), (rw, rb)) in enumerate(zip(items, refs)):
        assert rel_err(dw - 0.25 * (i + 1), 2 * rw) < 2e-4, i
    assert not ops.dw_group_supported(c, rows, 64, 128) and not ops.dw_group_supported(c, rows + 8, 128, 128)
    with pytest.raises(Exception):  # (a [64, 128] problem is legal: it runs on the 64-tile group, test_weight_gradient_ring_group_kernel)
        ops.linear_bwd_weight_group(c, [(rnd(rows, 96, dtype=ct), rnd(rows, 128, dtype=ct), torch.zeros(96, 128, device=DEV), None)])


@pytest.mark.parametrize("rows,C,nprob", [(147456, 64, 2), (4608, 64, 2), (2304, 64, 4), (64, 64, 2), (1152, 192, 3)])
def test_weight_gradient_ring_group_kernel(ops, rows, C, nprob):
    """focal_linear_bwd_weight_group on shapes the 128 x 128 tiles do not take (a 64-channel block: proj [C, C] and qkv [3C, C], plus fc2
    [C, 4C] / fc1 [4C, C] where the MLP branch is not fused): one launch of the 64 x 64 ring tiles behind a problem table =
    the single launches; accumulation into non-zero buffers, with and without bias, token ranges of one stage and of thousands."""
    ct = torch.bfloat16
    c = ops.code(ct)
    shapes = [(C, C), (3 * C, C), (C, 4 * C), (4 * C, C)][:nprob]
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        assert ops.dw_group_kind(c, rows, N, K) >= 1
        dy, x = rnd(rows, N, seed=400 + i, dtype=ct), rnd(rows, K, seed=410 + i, dtype=ct)
        dw = torch.full((N, K), 0.5 * (i + 1), device=DEV)
        db = torch.full((N,), -1.0, device=DEV) if i != 0 else None
        items.append((dy, x, dw, db))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    assert any(ops.dw_group_kind(c, rows, N, K) == 1 for N, K in shapes)
    wgs = ops.linear_bwd_weight_group_workgroups(c, items)
    tiles = sum((N // 64) * (K // 64) for N, K in shapes)
    assert tiles <= wgs <= max(256, tiles)
    ops.linear_bwd_weight_group(c, items)
    for i, ((dy, x, dw, db), (rw, rb)) in enumerate(zip(items, refs)):
        assert rel_err(dw - 0.5 * (i + 1), rw) < 2e-4, i
        if db is not None:
            assert rel_err(db + 1.0, rb) < 1e-4, i
    ops.linear_bwd_weight_group(c, items)  # accumulates
    for i, ((dy, x, dw, db), (rw, rb)) in enumerate(zip(items, refs)):
        assert rel_err(dw - 0.5 * (i + 1), 2 * rw) < 2e-4, i
    assert ops.dw_group_kind(c, rows, 128, 256) == 2 and ops.dw_group_kind(c, rows, 96, 64) == 0 and ops.dw_group_kind(c, rows + 8, 64, 64) == 0
    with pytest.raises(Exception):  # five 64-tile problems: more than the table holds
        ops.linear_bwd_weight_group(c, [items[0]] * 5)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_linear_bwd_fc2_chain(ops, ct):
    """fc2 of the Swin MLP: y = r + h W^T + b with fp32 dy -> du = (dy W) * h' and dW = dy^T h."""
    from focal_amd._lib import ACT_GELU, EPI_RESIDUAL
    M, N, K = 1152, 64, 256
    h, hg = rnd(M, K, seed=14, dtype=ct), rnd(M, K, seed=141, dtype=ct)
    w = rnd(N, K, scale=K ** -0.5, seed=15, dtype=ct)
    g = rnd(M, N, seed=16)
    c, f32 = ops.code(ct), ops.code(torch.float32)
    d = ops.linear_desc(c, M, N, K, c, f32, ACT_GELU, EPI_RESIDUAL)
    du = torch.empty(M, K, dtype=ct, device=DEV)
    ops.linear_bwd_data(d, g, w, hg, du)
    gq = g if ct == torch.float32 else g.bfloat16().float()
    assert rel_err(du.float(), (gq @ w.float()) * hg.float()) < (1e-5 if ct == torch.float32 else 8e-3)
    dw = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    ops.linear_bwd_weight(d, g, h, dw, db)
    assert rel_err(dw, gq.t() @ h.float()) < (2e-5 if ct == torch.float32 else 2e-4)
    assert rel_err(db, g.sum(0)) < 2e-5  # the bias gradient is summed in fp32 BEFORE the operand is rounded to bf16


def test_linear_relu_out_bwd(ops):
    ct = torch.float32
    B, E = 32, 256
    h = F.relu(rnd(B, E, seed=17))
    w, dz = rnd(E, E, scale=E ** -0.5, seed=18), rnd(B, E, seed=19)
    from focal_amd._lib import ACT_RELU_OUT
    f32 = ops.code(ct)
    d = ops.linear_desc(f32, B, E, E, f32, f32, ACT_RELU_OUT)
    dh = torch.empty(B, E, device=DEV)
    ops.linear_bwd_data(d, dz, w, h, dh)
    assert rel_err(dh, (dz @ w) * (h > 0)) < 1e-5


# The LDS-DMA pipelined bf16 kernel (gemm_pipe.hpp) takes over when K >= 128, K % 64 == 0, N % 64 == 0, M >= 1024; both tile
# shapes (N >= 384 and N % 128 == 0 -> 128 x 128, else 128 x 64), ragged M, every epilogue it implements, and the weight read
# in its [K][N] orientation for the data gradient.
@pytest.mark.parametrize("M,N,K", [(1100, 64, 128), (2048, 384, 128), (1333, 768, 256), (4608, 256, 1024), (1024, 192, 192)])
def test_linear_pipelined_kernel_fwd_and_data_gradient(ops, M, N, K):
    ct = torch.bfloat16
    x, w, b = rnd(M, K, seed=21, dtype=ct), rnd(N, K, scale=K ** -0.5, seed=22, dtype=ct), rnd(N, seed=23)
    y, _ = ops.linear(x, w, b, compute=ct)
    assert rel_err(y.float(), x.float() @ w.float().t() + b) < 6e-3
    yf, _ = ops.linear(x, w, b, compute=ct, y_dtype=torch.float32)
    assert rel_err(yf, x.float() @ w.float().t() + b) < 3e-3
    dy = rnd(M, N, seed=24, dtype=ct)
    c = ops.code(ct)
    d = ops.linear_desc(c, M, N, K, c, c)
    dx = torch.full((M, K), float("nan"), dtype=ct, device=DEV)
    ops.linear_bwd_data(d, dy, w, None, dx)
    assert rel_err(dx.float(), dy.float() @ w.float()) < 6e-3


# The persistent ring kernel (gemm_ring.hpp) takes the Swin stage-1 / stage-2 shapes from M >= 4096: every configuration the dispatcher
# picks (gemm_dispatch.inc: launch_ring; the LayerNorm epilogues in test_linear_resid_ln_wide / test_linear_bwd_data_ln below), full and
# ragged row counts (a last row tile with 8 valid rows), a workgroup walking one, several and unequal numbers of tiles.
@pytest.mark.parametrize("M", [4096, 9216 + 8, 18432])
def test_linear_ring_kernel_configurations(ops, M):
    from focal_amd import _lib
    from focal_amd._lib import ACT_GELU, EPI_GELU, EPI_RESIDUAL
    ct = torch.bfloat16
    c, f32 = ops.code(ct), ops.code(torch.float32)
    lib = _lib.load()

    def kernel():
        return lib.focal_last_kernel().decode()
    for C in (128, 256):
        # fc1 + GELU (+ derivative): wide output, panel resident (K = 128: 128 x 128 tiles; K = 256: 64 x 128)
        a, w1, b1 = rnd(M, C, seed=31, dtype=ct), rnd(4 * C, C, scale=C ** -0.5, seed=32, dtype=ct), rnd(4 * C, seed=33)
        hg = torch.full((M, 4 * C), float("nan"), dtype=ct, device=DEV)
        h, _ = ops.linear(a, w1, b1, compute=ct, epilogue=EPI_GELU, act_grad=hg)
        assert "focal_gemm_ring_kernel" in kernel(), kernel()
        u = (a.float() @ w1.float().t() + b1).requires_grad_(True)
        href = F.gelu(u)
        href.sum().backward()
        assert rel_err(h.float(), href) < 5e-3 and rel_err(hg.float(), u.grad) < 5e-3
        # dX of fc2 x the saved derivative: the weight read in its [K][N] orientation, aux rows requested before the k loop
        w2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=34, dtype=ct)
        gm = rnd(M, C, seed=35, dtype=ct)
        d2 = ops.linear_desc(c, M, C, 4 * C, c, c, ACT_GELU)
        du = torch.full((M, 4 * C), float("nan"), dtype=ct, device=DEV)
        ops.linear_bwd_data(d2, gm, w2, hg, du)
        assert "focal_gemm_ring_kernel" in kernel(), kernel()
        assert rel_err(du.float(), (gm.float() @ w2.float()) * hg.float()) < 8e-3
    C = 256
    # fc2 + residual (fp32 stream, K = 1024), proj + residual (K = 256), dX of fc1 / qkv (N = 256, K = 1024 / 768)
    for K in (1024, 256):
        x, w, b, r = rnd(M, K, seed=36, dtype=ct), rnd(C, K, scale=K ** -0.5, seed=37, dtype=ct), rnd(C, seed=38), rnd(M, C, seed=39)
        y, _ = ops.linear(x, w, b, compute=ct, y_dtype=torch.float32, resid=r, epilogue=EPI_RESIDUAL)
        assert "focal_gemm_ring_kernel" in kernel(), kernel()
        assert rel_err(y, r + x.float() @ w.float().t() + b) < 3e-3
    for N in (1024, 768):
        dy, w = rnd(M, N, seed=40, dtype=ct), rnd(N, C, scale=N ** -0.5, seed=41, dtype=ct)
        d = ops.linear_desc(c, M, N, C, c, c)
        dx = torch.full((M, C), float("nan"), dtype=ct, device=DEV)
        ops.linear_bwd_data(d, dy, w, None, dx)
        assert "focal_gemm_ring_kernel" in kernel(), kernel()
        assert rel_err(dx.float(), dy.float() @ w.float()) < 6e-3
    # qkv at stage 1 (K = 128, bf16 out)
    x, w, b = rnd(M, 128, seed=42, dtype=ct), rnd(384, 128, scale=128 ** -0.5, seed=43, dtype=ct), rnd(384, seed=44)
    y, _ = ops.linear(x, w, b, compute=ct)
    assert "focal_gemm_ring_kernel" in kernel(), kernel()
    assert rel_err(y.float(), x.float() @ w.float().t() + b) < 6e-3
    # ... and the shapes the lab left on the one-tile kernel stay there (qkv at K = 256)
    x, w, b = rnd(M, 256, seed=45, dtype=ct), rnd(768, 256, scale=256 ** -0.5, seed=46, dtype=ct), rnd(768, seed=47)
    y, _ = ops.linear(x, w, b, compute=ct)
    assert "focal_gemm_pipe_kernel" in kernel(), kernel()
    assert rel_err(y.float(), x.float() @ w.float().t() + b) < 6e-3


def test_linear_pipelined_kernel_epilogues(ops):
    from focal_amd._lib import ACT_GELU, EPI_GELU, EPI_RESIDUAL
    ct = torch.bfloat16
    M, C = 1300, 128
    a, w1, b1 = rnd(M, C, seed=31, dtype=ct), rnd(4 * C, C, scale=C ** -0.5, seed=32, dtype=ct), rnd(4 * C, seed=33)
    hg = torch.empty(M, 4 * C, dtype=ct, device=DEV)
    h, _ = ops.linear(a, w1, b1, compute=ct, epilogue=EPI_GELU, act_grad=hg)
    u = (a.float() @ w1.float().t() + b1).requires_grad_(True)
    href = F.gelu(u)
    href.sum().backward()
    assert rel_err(h.float(), href) < 5e-3 and rel_err(hg.float(), u.grad) < 5e-3
    w2, b2, r = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=34, dtype=ct), rnd(C, seed=35), rnd(M, C, seed=36)
    y, _ = ops.linear(h, w2, b2, compute=ct, y_dtype=torch.float32, resid=r, act_in=ACT_GELU, epilogue=EPI_RESIDUAL)
    assert rel_err(y, r + h.float() @ w2.float().t() + b2) < 4e-3
    # du = (gm W2) * h' with the pre-masked operand-dtype gradient (the Swin backward's fc2 data gradient)
    gm = rnd(M, C, seed=37, dtype=ct)
    c = ops.code(ct)
    d = ops.linear_desc(c, M, C, 4 * C, c, c, ACT_GELU)
    du = torch.empty(M, 4 * C, dtype=ct, device=DEV)
    ops.linear_bwd_data(d, gm, w2, hg, du)
    assert rel_err(du.float(), (gm.float() @ w2.float()) * hg.float()) < 8e-3


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K", [(1152, 64), (4610, 256)])
def test_linear_residual_with_fused_layernorm(ops, ct, M, K):
    """focal_linear_resid_ln_fwd = focal_linear_fwd (residual epilogue) + focal_layernorm_fwd of its output, one kernel."""
    from focal_amd._lib import ACT_NONE, EPI_RESIDUAL
    N = 64
    x, w, b = rnd(M, K, seed=41, dtype=ct), rnd(N, K, scale=K ** -0.5, seed=42, dtype=ct), rnd(N, seed=43)
    r, gamma, beta = rnd(M, N, seed=44), rnd(N, seed=45) * 0.2 + 1.0, rnd(N, seed=46) * 0.1
    c, f32 = ops.code(ct), ops.code(torch.float32)
    d = ops.linear_desc(c, M, N, K, c, f32, ACT_NONE, EPI_RESIDUAL)
    y = torch.full((M, N), float("nan"), device=DEV)
    y_ln, stats = ops.linear_resid_ln_fwd(d, x, w, b, r, y, gamma, beta, ct)
    y_ref, _ = ops.linear(x, w, b, compute=ct, y_dtype=torch.float32, resid=r, epilogue=EPI_RESIDUAL)
    assert torch.equal(y, y_ref)  # same kernel, same arithmetic
    ln_ref, st_ref = ops.layernorm_fwd(y_ref, gamma, beta, ct)
    assert rel_err(y_ln.float(), ln_ref.float()) < (1e-6 if ct == torch.float32 else 4e-3)
    assert rel_err(stats.view(-1), st_ref.view(-1)) < 1e-5
    ref = F.layer_norm(r + x.float() @ w.float().t() + b, (N,), gamma, beta, 1e-5)
    assert rel_err(y_ln.float(), ref) < (1e-5 if ct == torch.float32 else 8e-3)


# ---------------------------------------------------------------------------------------------- layer norm
@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(4608, 64), (1153, 128), (300, 256), (77, 512)])
def test_layernorm(ops, ct, rows, C):
    x, g, b = rnd(rows, C, scale=2.0, seed=20) + 0.5, 1 + 0.1 * rnd(C, seed=21), 0.1 * rnd(C, seed=22)
    y, stats = ops.layernorm_fwd(x, g, b, ct)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    assert rel_err(y.float(), ref) < (1e-5 if ct == torch.float32 else 4e-3)
    assert rel_err(stats[:, 0], x.mean(1)) < 1e-5
    dy = rnd(rows, C, seed=23, dtype=ct)
    ref.backward(dy.float())
    dx = rnd(rows, C, seed=24)
    base = dx.clone()
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.layernorm_bwd(dy, x, stats, g, dx, True, dg, db)
    assert rel_err(dx - base, xr.grad) < 2e-5
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_layernorm_patch_merge_gather(ops, ct):
    B, H, W, Cin = 3, 6, 24, 64
    x = rnd(B, H, W, Cin, seed=25)
    g, b = 1 + 0.1 * rnd(4 * Cin, seed=26), 0.1 * rnd(4 * Cin, seed=27)
    y, stats = ops.layernorm_fwd(x.view(-1, Cin), g, b, ct, gather=(B, H, W, Cin))
    xr = x.clone().requires_grad_(True)
    cat = torch.cat([xr[:, 0::2, 0::2], xr[:, 1::2, 0::2], xr[:, 0::2, 1::2], xr[:, 1::2, 1::2]], -1).view(-1, 4 * Cin)
    ref = F.layer_norm(cat, (4 * Cin,), g, b, 1e-5)
    assert rel_err(y.float(), ref) < (1e-5 if ct == torch.float32 else 4e-3)
    dy = rnd(y.shape[0], 4 * Cin, seed=28, dtype=ct)
    ref.backward(dy.float())
    dx = torch.full_like(x, 7.0)
    dg, db = torch.zeros(4 * Cin, device=DEV), torch.zeros(4 * Cin, device=DEV)
    ops.layernorm_bwd(dy, x.view(-1, Cin), stats, g, dx, False, dg, db, gather=(B, H, W, Cin))
    assert rel_err(dx, xr.grad) < 2e-5


# ---------------------------------------------------------------------------------------------- window attention
def _ref_window_attn(qkv, table, B, H, W, C, heads, wh, ww, sh, sw):
    from oracle.swt import _from_windows, _to_windows, relative_position_index, shifted_window_mask
    hd = C // heads
    x = qkv.view(B, H, W, 3 * C)
    shifted = sh > 0 and sw > 0
    if shifted:
        x = torch.roll(x, shifts=(-sh, -sw), dims=(1, 2))
    xw = _to_windows(x, wh, ww)
    Bw, N, _ = xw.shape
    q, k, v = xw.view(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    idx = relative_position_index(wh, ww).to(qkv.device)
    attn = attn + table[idx.reshape(-1)].view(N, N, heads).permute(2, 0, 1)[None]
    if shifted:
        mask = shifted_window_mask(H, W, wh, ww, sh, sw).to(qkv.device)
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(Bw, N, C)
    y = _from_windows(out, wh, ww, H, W)
    if shifted:
        y = torch.roll(y, shifts=(sh, sw), dims=(1, 2))
    return y.reshape(B * H * W, C)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W,C,sh,sw", [(12, 48, 64, 0, 0), (12, 48, 64, 1, 1), (6, 12, 128, 1, 1), (3, 6, 256, 0, 0), (3, 12, 256, 0, 1)])
def test_window_attention(ops, ct, H, W, C, sh, sw):
    B, heads, wh, ww = 5, 4, 3, 3
    M = B * H * W
    qkv = rnd(M, 3 * C, seed=30, dtype=ct)
    table = rnd(25, heads, scale=0.5, seed=31)
    c = ops.code(ct)
    d = ops.attn_desc(c, B, H, W, C, heads, wh, ww, sh, sw)
    out = torch.empty(M, C, dtype=ct, device=DEV)
    ops.window_attn_fwd(d, qkv, table, out)
    q32 = qkv.float().requires_grad_(True)
    t32 = table.clone().requires_grad_(True)
    ref = _ref_window_attn(q32, t32, B, H, W, C, heads, wh, ww, sh, sw)
    assert rel_err(out.float(), ref) < (2e-5 if ct == torch.float32 else 5e-3)
    do = rnd(M, C, seed=32, dtype=ct)
    ref.backward(do.float())
    dqkv = torch.empty_like(qkv)
    dt = torch.zeros_like(table)
    ops.window_attn_bwd(d, qkv, table, do, dqkv, dt)
    assert rel_err(dqkv.float(), q32.grad) < (3e-5 if ct == torch.float32 else 8e-3)
    assert rel_err(dt, t32.grad) < (1e-4 if ct == torch.float32 else 5e-3)

@pytest.mark.parametrize("H,W,sh,sw,p_attn", [(12, 48, 0, 0, 0.0), (12, 48, 1, 1, 0.0), (12, 24, 1, 1, 0.2), (12, 24, 0, 0, 0.2)])
def test_window_attention_with_qkv_projection_folded_in(ops, H, W, sh, sw, p_attn):
    """focal_window_attn_qkv_fwd / _bwd (64-channel blocks, bf16: the attention kernels project q / k / v of an item from the window's
    norm1 rows; reference models/SwinModules.py:113-152 + :294-334) against (1) the fp32 torch restatement `_ref_window_attn` applied
    to the fp32 qkv Linear (dropout off) and (2) the unfused HIP path -- focal_linear_fwd -> focal_window_attn_fwd / _bwd -- with the
    same seed words and stream id, i.e. the same attention-dropout mask (dropout on and off)."""
    B, C, heads, wh, ww = 5, 64, 4, 3, 3
    M = B * H * W
    ct = torch.bfloat16
    cc = ops.code(ct)
    assert ops.attn_qkv_supported(cc, C, heads, wh * ww) and not ops.attn_qkv_supported(cc, 128, heads, wh * ww)
    a1 = rnd(M, C, seed=130, dtype=ct)
    wqkv = rnd(3 * C, C, scale=C ** -0.5, seed=131, dtype=ct)
    bqkv = rnd(3 * C, scale=0.2, seed=132)
    table = rnd(25, heads, scale=0.5, seed=133)
    state = ops.new_rng_state(11, DEV) if p_attn > 0 else None
    d = ops.attn_desc(cc, B, H, W, C, heads, wh, ww, sh, sw, p_attn, state, 91)
    out = torch.empty(M, C, dtype=ct, device=DEV)
    ops.window_attn_qkv_fwd(d, a1, wqkv, bqkv, table, out)
    # unfused HIP path on the same operands
    d_qkv = ops.linear_desc(cc, M, 3 * C, C, cc, cc)
    qkv = torch.empty(M, 3 * C, dtype=ct, device=DEV)
    ops.linear_fwd(d_qkv, a1, wqkv, bqkv, None, qkv)
    out_u = torch.empty(M, C, dtype=ct, device=DEV)
    ops.window_attn_fwd(d, qkv, table, out_u)
    assert rel_err(out.float(), out_u.float()) < 4e-3, rel_err(out.float(), out_u.float())   # (q / k / v rounded to bf16 from two accumulation orders)
    do = rnd(M, C, seed=134, dtype=ct)
    dqkv, dt = torch.empty(M, 3 * C, dtype=ct, device=DEV), torch.zeros_like(table)
    ops.window_attn_qkv_bwd(d, a1, wqkv, bqkv, table, do, dqkv, dt)
    dqkv_u, dt_u = torch.empty_like(dqkv), torch.zeros_like(table)
    ops.window_attn_bwd(d, qkv, table, do, dqkv_u, dt_u)
    assert rel_err(dqkv.float(), dqkv_u.float()) < 6e-3 and rel_err(dt, dt_u) < 4e-3
    # ... and with the proj Linear's input gradient folded in as well: dout = dL/d(proj output), the kernel forms dout . Wproj per item
    wproj = rnd(C, C, scale=C ** -0.5, seed=135, dtype=ct)
    gy = rnd(M, C, seed=136, dtype=ct)
    d_proj = ops.linear_desc(cc, M, C, C, cc, cc)
    do_u = torch.empty(M, C, dtype=ct, device=DEV)
    ops.linear_bwd_data(d_proj, gy, wproj, None, do_u)
    dq_u, dtb_u = torch.empty_like(dqkv), torch.zeros_like(table)
    ops.window_attn_qkv_bwd(d, a1, wqkv, bqkv, table, do_u, dq_u, dtb_u)
    dq_f, dtb_f = torch.empty_like(dqkv), torch.zeros_like(table)
    ops.window_attn_qkv_bwd(d, a1, wqkv, bqkv, table, gy, dq_f, dtb_f, wproj=wproj)
    assert rel_err(dq_f.float(), dq_u.float()) < 2e-3 and rel_err(dtb_f, dtb_u) < 2e-3   # same products, same bf16 rounding of dO
    if p_attn == 0.0:  # ... and the reference formula in fp32
        a32, w32, b32 = a1.float().requires_grad_(True), wqkv.float(), bqkv.clone()
        t32 = table.clone().requires_grad_(True)
        q32 = a32 @ w32.t() + b32
        q32.retain_grad()
        ref = _ref_window_attn(q32, t32, B, H, W, C, heads, wh, ww, sh, sw)
        assert rel_err(out.float(), ref) < 6e-3
        ref.backward(do.float())
        assert rel_err(dqkv.float(), q32.grad) < 1e-2 and rel_err(dt, t32.grad) < 6e-3


# ---------------------------------------------------------------------------------------------- embed / fft
@pytest.mark.parametrize("S,pw,Hp,Wp,cin", [(1600, 40, 12, 48, 2), (20, 1, 12, 24, 2)])
def test_pad_patch_embed_ln(ops, S, pw, Hp, Wp, cin):
    B, I, C0 = 3, 10, 64
    x = rnd(B, cin, I, S, scale=20.0, seed=40)
    w, b = rnd(C0, cin, 1, pw, scale=(cin * pw) ** -0.5, seed=41), rnd(C0, seed=42)
    g, be = 1 + 0.1 * rnd(C0, seed=43), 0.1 * rnd(C0, seed=44)
    out = ops.pad_patch_embed_ln(x, w, b, g, be, Hp, Wp, pw)
    xp = F.pad(x, (0, Wp * pw - S, 0, Hp - I))
    ref = F.conv2d(xp, w, b, stride=(1, pw)).flatten(2).transpose(1, 2)
    ref = F.layer_norm(ref, (C0,), g, be, 1e-5).reshape(-1, C0)
    assert (out - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S,pw,Hp,Wp,cin", [(1600, 40, 12, 48, 2), (20, 1, 12, 24, 2)])
def test_pad_patch_embed_with_fused_next_layernorm(ops, ct, S, pw, Hp, Wp, cin):
    """focal_pad_patch_embed_ln2_fwd = the embedding kernel + focal_layernorm_fwd of its tokens (block 0's norm1)."""
    B, I, C0 = 3, 10, 64
    x = rnd(B, cin, I, S, seed=51)
    w, b = rnd(C0, cin, 1, pw, scale=(cin * pw) ** -0.5, seed=52), rnd(C0, seed=53) * 0.1
    g, be = rnd(C0, seed=54) * 0.1 + 1.0, rnd(C0, seed=55) * 0.1
    g2, be2 = rnd(C0, seed=56) * 0.2 + 1.0, rnd(C0, seed=57) * 0.1
    tok_ref = ops.pad_patch_embed_ln(x, w, b, g, be, Hp, Wp, pw)
    tok, y_ln, stats = ops.pad_patch_embed_ln(x, w, b, g, be, Hp, Wp, pw, next_ln=(g2, be2, ct))
    assert torch.equal(tok, tok_ref)
    ln_ref, st_ref = ops.layernorm_fwd(tok_ref, g2, be2, ct)
    assert rel_err(y_ln.float(), ln_ref.float()) < (1e-6 if ct == torch.float32 else 4e-3)
    assert rel_err(stats.view(-1), st_ref.view(-1)) < 1e-5


@pytest.mark.parametrize("n", [1600, 20, 64, 360])
def test_fft_realpack(ops, n):
    x = rnd(3, 2, 10, n, seed=50)
    out = ops.fft_realpack(x)
    f = torch.view_as_real(torch.fft.fft(x.double(), dim=-1))
    ref = f.permute(0, 1, 4, 2, 3).reshape(3, 4, 10, n)
    assert (out.double() - ref).abs().max().item() < 2e-4 * math.sqrt(n)


def test_fft_realpack_multi_equals_single_calls(ops):
    """focal_fft_realpack_multi: the transforms of a step's (view, modality) pairs in one call -- short rows (n <= 64, direct DFT) share
    launches of up to 8 problems, long rows go the usual way -- give exactly what the single calls give, augmentations included."""
    g = torch.Generator().manual_seed(70)
    xs = [torch.randn(5, 3, 10, 20, generator=g).to(DEV), torch.randn(5, 1, 10, 20, generator=g).to(DEV), torch.randn(4, 1, 10, 1600, generator=g).to(DEV),
          torch.randn(3, 2, 7, 48, generator=g).to(DEV), torch.randn(2, 1, 10, 64, generator=g).to(DEV)]
    kws = [dict(), dict(scale=-1.1), dict(flip=True), dict(perm=[3, 1, 0, 2, 6, 5, 4]), dict(phase=0.7)]
    items = [dict(x=x, **kw) for x, kw in zip(xs, kws)]
    items += [dict(x=xs[0], scale=0.9, perm=list(range(9, -1, -1))), dict(x=xs[1], flip=True, phase=-1.3)] + [dict(x=xs[0], scale=float(i)) for i in range(2, 9)]
    both = torch.empty(10, 6, 10, 20, device=DEV)
    items += [dict(x=xs[0], out=both[:5]), dict(x=xs[0], scale=-1.1, out=both[5:])]
    outs = ops.fft_realpack_multi(items)
    assert len(outs) == len(items) and outs[-1].data_ptr() == both[5:].data_ptr()
    for it, o in zip(items, outs):
        ref = ops.fft_realpack(**{k: v for k, v in it.items() if k != "out"})
        assert o.shape == ref.shape
        tol = 1e-5 * ref.abs().max().item()
        assert (o - ref).abs().max().item() <= tol, (it["x"].shape, {k: v for k, v in it.items() if k not in ("x", "out")})
    f = torch.view_as_real(torch.fft.fft(xs[0].double(), dim=-1)).permute(0, 1, 4, 2, 3).reshape(5, 6, 10, 20)
    assert (outs[0].double() - f).abs().max().item() < 2e-4 * math.sqrt(20)


# ---------------------------------------------------------------------------------------------- optimizer / rng
def test_adamw_matches_torch(ops):
    n = 4096 + 8
    p0, g = rnd(n, seed=60), rnd(n, seed=61)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=0.05)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    state = ops.new_rng_state(123, DEV)
    lr = torch.full((1,), 1e-3, device=DEV)
    for it in range(3):
        pt.grad = g * (it + 1)
        opt.step()
        ops.rng_advance(state)
        ops.adamw_multi([(p, g * (it + 1), m, v, shadow)], lr, state)
    assert (p - pt.detach()).abs().max().item() < 1e-6
    assert torch.equal(shadow, p.bfloat16())
    assert int(state[1]) == 3


def test_adamw_advances_its_own_step_count(ops):
    """focal_adamw_multi_advance: the update with step count state[1] + 1, then the last workgroup advances the step state and the dropout
    seed state exactly as focal_rng_advance would have -- several segments (the last one carries the advance), a large one (2048
    workgroups racing for the ticket) and a replayed hipGraph."""
    n0, n1 = 2048 * 1024 + 64, 4096
    p0, g0, p1, g1 = rnd(n0, seed=62), rnd(n0, seed=63), rnd(n1, seed=64), rnd(n1, seed=65)
    lr = torch.full((1,), 1e-3, device=DEV)

    def run(advance, steps=3, graph=False):
        segs = [(p0.clone(), g0, torch.zeros(n0, device=DEV), torch.zeros(n0, device=DEV), torch.zeros(n0, dtype=torch.bfloat16, device=DEV)),
                (p1.clone(), g1, torch.zeros(n1, device=DEV), torch.zeros(n1, device=DEV), torch.zeros(n1, dtype=torch.bfloat16, device=DEV))]
        step, seed = ops.new_step_state(DEV), ops.new_rng_state(99, DEV)

        def one():
            if advance:
                ops.adamw_multi(segs, lr, step, advance=True, seed_state=seed)
            else:
                ops.rng_advance(step)
                ops.rng_advance(seed)
                ops.adamw_multi(segs, lr, step)
        if graph:
            one()
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    one()
            for _ in range(steps - 1):  # (the capture itself runs nothing)
                gr.replay()
        else:
            for _ in range(steps):
                one()
        torch.cuda.synchronize()
        return segs, step, seed
    ref, rstep, rseed = run(False)
    for graph in (False, True):
        got, step, seed = run(True, graph=graph)
        assert torch.equal(step.cpu()[:2], rstep.cpu()[:2]) and int(step[1]) == 3 and int(step[2:].abs().sum()) == 0
        assert torch.equal(seed.cpu(), rseed.cpu())
        for a, b in zip(got, ref):
            assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_dropout_statistics_and_replay(ops):
    """Dropout / DropPath masks: right keep-rate, 1/(1-p) scaling, identical between the forward epilogue and the
    backward loader, and different after the seed advances."""
    from focal_amd._lib import EPI_RESIDUAL
    M, N, K, L = 4096, 64, 64, 64
    state = ops.new_rng_state(7, DEV)
    x = torch.ones(M, K, device=DEV)
    w = torch.eye(N, K, device=DEV)
    r = torch.zeros(M, N, device=DEV)
    dd = ops.drop_desc(state, 11, 0.2, 12, 0.1, L)
    y, d = ops.linear(x, w, None, compute=torch.float32, y_dtype=torch.float32, resid=r, epilogue=EPI_RESIDUAL, out_drop=dd)
    vals = y.unique()
    assert len(vals) == 2 and vals[0] == 0 and abs(vals[1].item() - 1 / (0.8 * 0.9)) < 1e-5
    keep_rows = (y.view(M // L, L * N).abs().sum(1) > 0).float().mean().item()
    assert abs(keep_rows - 0.9) < 0.12
    kept = y.view(M // L, L * N)[y.view(M // L, L * N).abs().sum(1) > 0]
    assert abs((kept > 0).float().mean().item() - 0.8) < 0.01
    # backward sees the same mask: dx = (dy * mask) @ w with w = I
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd_data(d, torch.ones(M, N, device=DEV), w, None, dx)
    assert torch.equal(dx > 0, y > 0)
    ops.rng_advance(state)
    y2, _ = ops.linear(x, w, None, compute=torch.float32, y_dtype=torch.float32, resid=r, epilogue=EPI_RESIDUAL, out_drop=dd)
    assert not torch.equal(y2 > 0, y > 0)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4096, 256, 64), (4096, 512, 128)])
def test_gelu_epilogue_dropout_rate_scale_and_consistency(ops, ct, M, N, K):
    """Mlp.drop inside the fc1 epilogue (both GEMM kernels: 64 x 64 at K = 64, pipelined at K = 128): keep rate 1 - p,
    survivors scaled by 1 / (1 - p), the saved derivative carries exactly the same mask, a new mask after the seed advances.
    (This site draws one hash per PAIR of elements: 16-bit thresholds.)"""
    from focal_amd._lib import EPI_GELU
    p = 0.2
    state = ops.new_rng_state(11, DEV)
    a = (torch.rand(M, K, device=DEV) + 0.5).to(ct)          # positive inputs, positive weights: gelu(u) > 0, gelu'(u) > 0
    w = (torch.rand(N, K, device=DEV) * K ** -0.5 + 0.05).to(ct)
    dd = ops.drop_desc(state, 21, p, 0, 0.0, 1)
    hg = torch.empty(M, N, dtype=ct, device=DEV)
    h, _ = ops.linear(a, w, None, compute=ct, epilogue=EPI_GELU, act_grad=hg, out_drop=dd)
    hg0 = torch.empty(M, N, dtype=ct, device=DEV)
    h0, _ = ops.linear(a, w, None, compute=ct, epilogue=EPI_GELU, act_grad=hg0)
    kept = h != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 4e-3
    assert torch.equal(kept, hg != 0)
    assert rel_err(h[kept].float(), h0[kept].float() / (1 - p)) < (1e-6 if ct == torch.float32 else 8e-3)
    assert rel_err(hg[kept].float(), hg0[kept].float() / (1 - p)) < (1e-6 if ct == torch.float32 else 8e-3)
    # neighbours are independent: P(both of a pair kept) = (1 - p)^2
    both = (kept[:, 0::2] & kept[:, 1::2]).float().mean().item()
    assert abs(both - (1 - p) ** 2) < 5e-3
    ops.rng_advance(state)
    h2, _ = ops.linear(a, w, None, compute=ct, epilogue=EPI_GELU, act_grad=hg, out_drop=dd)
    assert not torch.equal(h2 != 0, kept)


# ---------------------------------------------------------------------------------------------- DeepSense pieces
@pytest.mark.parametrize("S_in,k,stride,pad", [(1600, 80, 80, 0), (20, 3, 1, 1)])
def test_conv_in(ops, S_in, k, stride, pad):
    B, cin, I, C = 3, 2, 10, 64
    S_out = (S_in + 2 * pad - k) // stride + 1
    x = rnd(B, cin, I, S_in, scale=10.0, seed=70)
    w, b = rnd(C, cin, 1, k, scale=(cin * k) ** -0.5, seed=71), rnd(C, seed=72)
    d = ops.conv_in_desc(B, cin, I, S_in, S_out, k, stride, pad, C)
    z = ops.conv_in_fwd(d, x, w, b)
    ref = F.conv2d(x, w, b, stride=(1, stride), padding=(0, pad))  # [B, C, I, S_out]
    ref_tok = ref.permute(0, 2, 3, 1).reshape(-1, C)
    assert rel_err(z, ref_tok) < 1e-5
    dz = rnd(B * I * S_out, C, seed=73)
    dw, db = torch.zeros_like(w), torch.zeros(C, device=DEV)
    ops.conv_in_bwd_weight(d, x, dz, dw, db)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    out = F.conv2d(x, wr, br, stride=(1, stride), padding=(0, pad)).permute(0, 2, 3, 1).reshape(-1, C)
    (out * dz).sum().backward()
    assert rel_err(dw, wr.grad) < 2e-5 and rel_err(db, br.grad) < 2e-5


@pytest.mark.parametrize("B,I,S", [(3, 10, 20), (64, 10, 80), (7, 10, 9)])
def test_conv_fwd_with_batchnorm_statistics_in_the_epilogue(ops, B, I, S):
    """focal_conv_fwd_bn == focal_conv_fwd followed by focal_bn_stats (training mode): z bit for bit, mean / rstd and the running buffers
    to summation order; 1 / 800 / a ragged number of workgroups (the last-arriver finalisation, 16 slots of partial sums)."""
    C, k, ct = 64, 3, torch.bfloat16
    rows = B * I * S
    x = rnd(rows, C, seed=174, dtype=ct)
    w, b = rnd(C, C, 1, k, scale=(C * k) ** -0.5, seed=175), rnd(C, seed=176)
    d = ops.conv_desc(ops.code(ct), rows, S, C, C, k)
    w_fwd = ops.permute_pack(w, C, C, k, ct)
    d_bn = ops.bn_desc(ops.code(ct), rows, C, I * S, 0.0, None, 0, momentum=0.1)
    rm0, rv0 = rnd(C, seed=177), rnd(C, seed=178).abs() + 0.5
    z_ref = ops.conv_fwd(d, x, w_fwd, b)
    rm, rv = rm0.clone(), rv0.clone()
    mr_ref = ops.bn_stats(d_bn, z_ref, rm, rv, True)
    rm2, rv2 = rm0.clone(), rv0.clone()
    z, mr = ops.conv_fwd_bn(d, x, w_fwd, b, d_bn, rm2, rv2)
    assert torch.equal(z, z_ref)
    assert rel_err(mr, mr_ref) < 1e-5 and rel_err(rm2, rm) < 1e-5 and rel_err(rv2, rv) < 1e-5
    # and against the definition
    mean, var = z_ref.mean(0), z_ref.var(0, unbiased=False)
    assert rel_err(mr[:C], mean) < 1e-4 and rel_err(mr[C:], (var + d_bn.eps).rsqrt()) < 1e-4


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_weight_gradients_with_fp32_operands_as_one_launch(ops, ct):
    """focal_linear_bwd_weight_group_f32 (round 5: the GRU's W_hh / W_ih gradients of a DeepSense pass, fp32 dy and x): eight problems of
    different shapes in one launch == the single launches == dy^T x, accumulated onto what the buffers hold; bias gradients where asked."""
    shapes = [(5120, 768, 256), (5120, 768, 128), (5120, 768, 512), (2560, 384, 128), (1000, 96, 40), (5120, 768, 256), (640, 64, 64), (5120, 128, 1600)]
    items, refs, singles = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(M, N, seed=300 + i), rnd(M, K, seed=320 + i)
        dw0 = rnd(N, K, seed=340 + i)
        db0 = rnd(N, seed=360 + i) if i % 2 == 0 else None
        dw, db = dw0.clone(), (db0.clone() if db0 is not None else None)
        items.append((dy, x, dw, db))
        a, b = (dy, x) if ct == torch.float32 else (dy.bfloat16().float(), x.bfloat16().float())
        refs.append((dw0.double() + a.double().t() @ b.double(), None if db0 is None else db0.double() + a.double().sum(0)))
        dw1, db1 = dw0.clone(), (db0.clone() if db0 is not None else None)
        d = ops.linear_desc(ops.code(ct), M, N, K, ops.code(torch.float32), ops.code(torch.float32))
        ops.linear_bwd_weight(d, dy, x, dw1, db1)
        singles.append((dw1, db1))
    ops.linear_bwd_weight_group_f32(ops.code(ct), items)
    # (bf16: the reference rounds the operands the same way; what is left is fp32 accumulation order over up to 5 120 rows)
    tol_ref, tol_single = (2e-5, 2e-5) if ct == torch.float32 else (1e-3, 2e-4)
    for (dy, x, dw, db), (rw, rb), (sw, sb) in zip(items, refs, singles):
        assert rel_err(dw.double(), rw) < tol_ref and rel_err(dw, sw) < tol_single
        if db is not None:
            # (the bias gradient sums dy as it was loaded: in bf16 mode the reference's operand rounding is the difference)
            assert rel_err(db.double(), rb) < (tol_ref if ct == torch.float32 else 5e-3) and rel_err(db, sb) < tol_single
    with pytest.raises(Exception, match="1 .. 8 problems"):
        ops.linear_bwd_weight_group_f32(ops.code(ct), items + items[:1])


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,p_drop", [(4, 0.0), (6, 0.3)])
def test_batchnorm_statistic_groups_equal_separate_calls(ops, ct, B, p_drop):
    """focal_bn_desc.groups = 2 (round 5: the two views of a step as one batch of 2B windows, each normalised by its own batch statistics):
    statistics, running-buffer sinks, forward, backward and the parameter gradients equal two one-group calls on the halves.  With
    Dropout2d on, the samples are numbered through the whole tensor: the second half equals a call whose rows start at sample B."""
    I, S, C = 10, 20, 64
    rows = B * I * S
    z = rnd(2 * rows, C, scale=2.0, seed=280) + 0.3
    z[rows:] = z[rows:] * 1.7 - 0.9          # the halves have different statistics
    gam, bet, res = 1 + 0.1 * rnd(C, seed=281), 0.1 * rnd(C, seed=282), rnd(2 * rows, C, seed=283)
    g = rnd(2 * rows, C, seed=284)
    state = ops.new_rng_state(9, DEV) if p_drop > 0 else None
    d2 = ops.bn_desc(ops.code(ct), 2 * rows, C, I * S, p_drop, state, 40, momentum=1.0, groups=2)
    sink_m, sink_v = torch.zeros(2, C, device=DEV), torch.zeros(2, C, device=DEV)
    mr2 = ops.bn_stats(d2, z, sink_m, sink_v, True)
    y2, ya2 = ops.bn_act_fwd(d2, z, mr2, gam, bet, res, ct)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz2 = ops.bn_act_bwd(d2, z, g, mr2, gam, bet, dg2, db2, ct)
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    for h in range(2):
        sl = slice(h * rows, (h + 1) * rows)
        if p_drop > 0 and h == 1:
            # a one-group call numbers its samples from 0: give it the whole tensor's statistics rows through a descriptor over 2B samples
            # whose first half is a copy of the second (same statistics), and compare the rows of its second half
            zz, gg, rr = torch.cat([z[sl], z[sl]]), torch.cat([g[sl], g[sl]]), torch.cat([res[sl], res[sl]])
            dd = ops.bn_desc(ops.code(ct), 2 * rows, C, I * S, p_drop, state, 40, momentum=1.0)
            rm, rv = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            mr = ops.bn_stats(dd, zz, rm, rv, True)
            y, ya = ops.bn_act_fwd(dd, zz, mr, gam, bet, rr, ct)
            assert rel_err(mr2[2 * C:], mr) < 1e-5
            assert rel_err(y2[sl], y[rows:]) < 1e-5 and rel_err(ya2[sl].float(), ya[rows:].float()) < (1e-5 if ct == torch.float32 else 2e-3)
            continue
        d1 = ops.bn_desc(ops.code(ct), rows, C, I * S, p_drop, state, 40, momentum=1.0)
        rm, rv = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        mr = ops.bn_stats(d1, z[sl], rm, rv, True)
        # (the column sums are atomic adds: equal to summation order, not bit for bit)
        assert rel_err(mr2[h * 2 * C:(h + 1) * 2 * C], mr) < 1e-6 and rel_err(sink_m[h], rm) < 1e-6 and rel_err(sink_v[h], rv) < 1e-6
        y, ya = ops.bn_act_fwd(d1, z[sl], mr, gam, bet, res[sl], ct)
        assert rel_err(y2[sl], y) < 1e-5 and rel_err(ya2[sl].float(), ya.float()) < (1e-5 if ct == torch.float32 else 2e-3)
        dz = ops.bn_act_bwd(d1, z[sl], g[sl], mr, gam, bet, dg1, db1, ct)
        assert rel_err(dz2[sl].float(), dz.float()) < (1e-5 if ct == torch.float32 else 2e-3)
    if p_drop == 0:
        assert rel_err(dg2, dg1) < 1e-5 and rel_err(db2, db1) < 1e-5
    # the two recorded statistics become the reference's two running-buffer updates (view 1, then view 2)
    run = rnd(C, seed=285)
    want = 0.9 * (0.9 * run + 0.1 * sink_m[0]) + 0.1 * sink_m[1]
    ops.bn_running_combine([run], [sink_m[0]], [sink_m[1]], 0.1)
    assert rel_err(run, want) < 1e-6


def test_conv_fwd_batchnorm_epilogue_with_statistic_groups(ops):
    """focal_conv_fwd_bn with two statistic groups == the one-group call on each half (rows per group a multiple of the tile height)."""
    B, I, S, C, k, ct = 32, 10, 20, 64, 3, torch.bfloat16
    rows = B * I * S                       # 6400 = 50 x 128
    x = rnd(2 * rows, C, seed=290, dtype=ct)
    x[rows:] = (x[rows:].float() * 1.5 + 0.25).to(ct)
    w, b = rnd(C, C, 1, k, scale=(C * k) ** -0.5, seed=291), rnd(C, seed=292)
    w_fwd = ops.permute_pack(w, C, C, k, ct)
    d2 = ops.conv_desc(ops.code(ct), 2 * rows, S, C, C, k)
    bn2 = ops.bn_desc(ops.code(ct), 2 * rows, C, I * S, 0.0, None, 0, momentum=1.0, groups=2)
    sm, sv_ = torch.zeros(2, C, device=DEV), torch.zeros(2, C, device=DEV)
    z2, mr2 = ops.conv_fwd_bn(d2, x, w_fwd, b, bn2, sm, sv_)
    d1 = ops.conv_desc(ops.code(ct), rows, S, C, C, k)
    bn1 = ops.bn_desc(ops.code(ct), rows, C, I * S, 0.0, None, 0, momentum=1.0)
    for h in range(2):
        rm, rv = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        z, mr = ops.conv_fwd_bn(d1, x[h * rows:(h + 1) * rows].contiguous(), w_fwd, b, bn1, rm, rv)
        assert torch.equal(z2[h * rows:(h + 1) * rows], z)
        assert rel_err(mr2[h * 2 * C:(h + 1) * 2 * C], mr) < 1e-5 and rel_err(sm[h], rm) < 1e-5 and rel_err(sv_[h], rv) < 1e-5
    bad = ops.bn_desc(ops.code(ct), 2 * 1620, C, I * S, 0.0, None, 0, momentum=1.0, groups=2)   # 1620 rows per group: not whole tiles
    with pytest.raises(Exception, match="multiple of"):
        ops.conv_fwd_bn(ops.conv_desc(ops.code(ct), 3240, S, C, C, k), x[:3240].contiguous(), w_fwd, b, bad, sm, sv_)


@pytest.mark.parametrize("B,I,S,k,groups", [(16, 10, 20, 5, 1), (32, 10, 20, 3, 2), (64, 10, 9, 5, 1), (512, 10, 20, 5, 2), (512, 10, 20, 3, 1), (4, 4, 4, 3, 1)])
def test_conv1xk_row_ring_kernel(ops, B, I, S, k, groups, monkeypatch):
    """conv_ring_kernel (64 -> 64 channels, bf16, whole 64-row tiles: the DeepSense step's inter-layer convolutions and their data gradients)
    against the definition in fp64 on the same bf16 operands, against the sliding-window GEMM it replaces (FOCAL_CONV_RING=0), and its BatchNorm
    statistics against the tensor it wrote; one tile per workgroup, runs of 2 and 4 tiles, intervals shorter than a fragment, two statistic groups."""
    C, ct = 64, torch.bfloat16
    rows = B * I * S
    assert rows % (64 * groups) == 0
    x = rnd(rows, C, seed=374, dtype=ct)
    w, b = rnd(C, C, 1, k, scale=(C * k) ** -0.5, seed=375), rnd(C, seed=376)
    d = ops.conv_desc(ops.code(ct), rows, S, C, C, k)
    w_fwd, w_bwd = ops.permute_pack(w, C, C, k, ct), ops.conv_pack_bwd(d, w, ct)
    dz, g_in = rnd(rows, C, seed=377, dtype=ct), rnd(rows, C, seed=378)
    d_bn = ops.bn_desc(ops.code(ct), rows, C, I * S, 0.0, None, 0, momentum=1.0, groups=groups)

    from focal_amd import _lib
    lib = _lib.load()

    def run():
        g = g_in.clone()
        torch.cuda.synchronize()
        _lib.check(lib.focal_trace_begin(64, _lib.TRACE_DISPATCH))
        try:
            z = ops.conv_fwd(d, x, w_fwd, b)
            rm, rv = torch.zeros(groups, C, device=DEV), torch.zeros(groups, C, device=DEV)
            z2, mr = ops.conv_fwd_bn(d, x, w_fwd, b, d_bn, rm, rv)
            ops.conv_bwd_data(d, dz, w_bwd, g, g)
            torch.cuda.synchronize()
        finally:
            lib.focal_trace_end()
        n = lib.focal_trace_count()
        recs = (_lib.TraceRecord * max(n, 1))()
        _lib.check(lib.focal_trace_read(0, n, recs))
        return z, z2, mr, rm, rv, g, [recs[i].kernel.decode() for i in range(n)]

    monkeypatch.setenv("FOCAL_CONV_RING", "0")
    z0, z20, mr0, rm0, rv0, g0, names0 = run()
    assert not any("conv_ring_kernel" in n for n in names0), names0
    monkeypatch.delenv("FOCAL_CONV_RING")
    z, z2, mr, rm, rv, g, names = run()
    assert sum("conv_ring_kernel" in n for n in names) == 3, names   # (the default: csrc/conv_ring.hpp)
    # the definition, in fp64 on the bf16-rounded operands
    wq = w.bfloat16().double()
    xi = x.double().view(B * I, S, C).permute(0, 2, 1).unsqueeze(2).requires_grad_(True)
    ref = F.conv2d(xi, wq, b.double(), padding=(0, k // 2)).squeeze(2).permute(0, 2, 1).reshape(rows, C)
    assert rel_err(z, ref.float()) < 2e-6 and torch.equal(z2, z)
    assert rel_err(z, z0) < 2e-6 and rel_err(g, g0) < 2e-6
    ref.backward(dz.double())
    dx_ref = xi.grad.squeeze(2).permute(0, 2, 1).reshape(rows, C)
    assert rel_err(g - g_in, dx_ref.float()) < 2e-5
    rg = rows // groups
    for h in range(groups):
        zh = z[h * rg:(h + 1) * rg].double()
        mean, var = zh.mean(0), zh.var(0, unbiased=False)
        assert rel_err(mr[h * 2 * C:h * 2 * C + C], mean.float()) < 1e-4 and rel_err(mr[h * 2 * C + C:(h + 1) * 2 * C], (var + d_bn.eps).rsqrt().float()) < 1e-4
        assert rel_err(rm[h], mean.float()) < 1e-4 and rel_err(rv[h], (var * rg / (rg - 1)).float()) < 1e-4
    assert rel_err(mr, mr0) < 1e-5 and rel_err(rm, rm0) < 1e-5 and rel_err(rv, rv0) < 1e-5
    # the sums-only form + the BatchNorm launch that finishes the statistics == the one-launch form followed by focal_bn_act_fwd
    assert ops.conv_fwd_bn_sums_supported(d, d_bn, x, w_fwd)
    gam, bet, res = 1 + 0.1 * rnd(C, seed=379), 0.1 * rnd(C, seed=380), rnd(rows, C, seed=381)
    y0, ya0 = ops.bn_act_fwd(d_bn, z2, mr, gam, bet, res, ct)
    zs, sums = ops.conv_fwd_bn_sums(d, x, w_fwd, b, d_bn)
    rm1, rv1 = torch.zeros(groups, C, device=DEV), torch.zeros(groups, C, device=DEV)
    y1, ya1, mr1 = ops.bn_act_fwd_sums(d_bn, zs, sums, rm1, rv1, gam, bet, res, ct)
    assert torch.equal(zs, z2) and rel_err(mr1, mr) < 1e-6 and rel_err(rm1, rm) < 1e-6 and rel_err(rv1, rv) < 1e-6
    # (the slot sums of two launches differ in their last bits -- atomics in another order --, so a few bf16 copies round the other way)
    assert rel_err(y1, y0) < 1e-5 and (ya1 != ya0).float().mean().item() < 1e-3 and rel_err(ya1.float(), ya0.float()) < 1e-2
    monkeypatch.setenv("FOCAL_CONV_BN_SUMS", "0")
    assert not ops.conv_fwd_bn_sums_supported(d, d_bn, x, w_fwd)


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [3, 5])
def test_conv1xk_same_as_sliding_window_gemm(ops, ct, k):
    B, I, S, C = 3, 10, 20, 64
    rows = B * I * S
    x = rnd(rows, C, seed=74, dtype=ct)
    w, b = rnd(C, C, 1, k, scale=(C * k) ** -0.5, seed=75), rnd(C, seed=76)
    d = ops.conv_desc(ops.code(ct), rows, S, C, C, k)
    w_fwd = ops.permute_pack(w, C, C, k, ct)
    w_bwd = ops.conv_pack_bwd(d, w, ct)
    z = ops.conv_fwd(d, x, w_fwd, b)
    wq = w if ct == torch.float32 else w.bfloat16().float()
    xi = x.float().view(B * I, S, C).permute(0, 2, 1).unsqueeze(2).requires_grad_(True)  # [B*I, C, 1, S]
    wr = wq.clone().requires_grad_(True)
    ref4 = F.conv2d(xi, wr, b, padding=(0, k // 2))
    ref = ref4.squeeze(2).permute(0, 2, 1).reshape(rows, C)
    assert rel_err(z, ref) < (1e-5 if ct == torch.float32 else 5e-3)
    dz = rnd(rows, C, seed=77, dtype=ct)
    ref.backward(dz.float())
    g_in = rnd(rows, C, seed=78)
    g = g_in.clone()
    ops.conv_bwd_data(d, dz, w_bwd, g, g)
    dx_ref = xi.grad.squeeze(2).permute(0, 2, 1).reshape(rows, C)
    assert rel_err(g - g_in, dx_ref) < (1e-5 if ct == torch.float32 else 6e-3)
    dwp = torch.zeros(C, k * C, device=DEV)
    db = torch.zeros(C, device=DEV)
    ops.conv_bwd_weight(d, dz, x, dwp, db)
    dw = torch.zeros_like(w)
    ops.permute_unpack_add(dwp, dw, C, C, k)
    assert rel_err(dw, wr.grad) < (2e-5 if ct == torch.float32 else 2e-4)
    assert rel_err(db, dz.float().sum(0)) < 1e-4


@pytest.mark.parametrize("ct", [torch.float32, torch.bfloat16])
def test_batchnorm_gelu_residual(ops, ct):
    B, I, S, C = 4, 10, 20, 64
    rows = B * I * S
    z = rnd(rows, C, scale=3.0, seed=80) + 0.7
    gam, bet, res = 1 + 0.1 * rnd(C, seed=81), 0.1 * rnd(C, seed=82), rnd(rows, C, seed=83)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    d = ops.bn_desc(ops.code(ct), rows, C, I * S)
    mr = ops.bn_stats(d, z, rm, rv, True)
    y, ya = ops.bn_act_fwd(d, z, mr, gam, bet, res, ct)
    zr = z.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.1).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet)
    ref = res + F.gelu(F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5))
    bn(z)
    assert rel_err(y, ref) < 2e-5
    assert rel_err(rm, bn.running_mean) < 1e-4 and rel_err(rv, bn.running_var) < 1e-4
    assert rel_err(ya.float(), ref) < (2e-5 if ct == torch.float32 else 4e-3)
    g = rnd(rows, C, seed=84)
    ref.backward(g)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz = ops.bn_act_bwd(d, z, g, mr, gam, bet, dg, db, ct)
    assert rel_err(dz.float(), zr.grad) < (5e-5 if ct == torch.float32 else 5e-3)
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    # eval mode: statistics come from the running buffers
    mr_e = ops.bn_stats(d, z, rm, rv, False)
    assert rel_err(mr_e[:C], rm) < 1e-6 and rel_err(mr_e[C:], torch.rsqrt(rv + 1e-5)) < 1e-5


def test_window_attention_dropout_fwd_bwd_consistent(ops):
    """Attention dropout: forward and backward regenerate the same mask (directional derivative check, exact fp32
    kernel), and the bf16 matrix-core kernel draws the same mask as the fp32 VALU kernel."""
    import os
    B, H, W, C, heads = 3, 6, 12, 64, 4
    M = B * H * W
    state = ops.new_rng_state(5, DEV)
    table = rnd(25, heads, scale=0.5, seed=91)
    qkv = rnd(M, 3 * C, seed=90)
    do = rnd(M, C, seed=92)
    d = ops.attn_desc(ops.code(torch.float32), B, H, W, C, heads, 3, 3, 1, 1, 0.2, state, 77)

    def f(x):
        o = torch.empty(M, C, device=DEV)
        ops.window_attn_fwd(d, x, table, o)
        return o
    dqkv, dt = torch.empty_like(qkv), torch.zeros_like(table)
    ops.window_attn_bwd(d, qkv, table, do, dqkv, dt)
    v = rnd(M, 3 * C, seed=93)
    eps = 1e-2
    num = ((f(qkv + eps * v) - f(qkv - eps * v)) * do).sum().item() / (2 * eps)
    ana = (dqkv * v).sum().item()
    assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (num, ana)
    # same mask in the bf16 MFMA kernel: outputs agree to bf16 rounding, which a different mask would not
    d16 = ops.attn_desc(ops.code(torch.bfloat16), B, H, W, C, heads, 3, 3, 1, 1, 0.2, state, 77)
    o16 = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
    ops.window_attn_fwd(d16, qkv.bfloat16(), table, o16)
    assert rel_err(o16.float(), f(qkv.bfloat16().float())) < 1e-2
    # ... and the MFMA backward against the exact-fp32 kernel on the bf16-rounded operands (same seed words, same stream id -> same mask)
    q16, do16 = qkv.bfloat16().float(), do.bfloat16().float()
    g1, g2 = torch.empty(M, 3 * C, device=DEV), torch.zeros_like(table)
    ops.window_attn_bwd(d, q16, table, do16, g1, g2)
    h1, h2 = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=DEV), torch.zeros_like(table)
    ops.window_attn_bwd(d16, qkv.bfloat16(), table, do.bfloat16(), h1, h2)
    assert rel_err(h1.float(), g1) < 2e-2 and rel_err(h2, g2) < 2e-2


def test_masked_gradient_copy_matches_forward_mask(ops):
    """layernorm_bwd's dx_masked / mask_cast output = completed dx * (the mask the forward residual epilogue drew), in
    both row layouts (plain and PatchMerging gather)."""
    from focal_amd._lib import EPI_RESIDUAL
    B, H, W, Cin = 4, 6, 12, 64
    M, L = B * H * W, H * W
    state = ops.new_rng_state(9, DEV)
    dd = ops.drop_desc(state, 21, 0.2, 25, 0.1, L)
    # forward mask through the residual epilogue: y = 0 + mask * (1 @ I)
    y, _ = ops.linear(torch.ones(M, Cin, device=DEV), torch.eye(Cin, device=DEV), None, compute=torch.float32, y_dtype=torch.float32,
                      resid=torch.zeros(M, Cin, device=DEV), epilogue=EPI_RESIDUAL, out_drop=dd)
    g = rnd(M, Cin, seed=5)
    assert torch.equal(ops.mask_cast(g, dd, torch.float32), g * y)
    assert rel_err(ops.mask_cast(g, dd, torch.bfloat16).float(), g * y) < 5e-3
    # plain LayerNorm backward, accumulate into an existing gradient
    x, gam = rnd(M, Cin, seed=1), rnd(Cin, seed=2)
    _, st = ops.layernorm_fwd(x, gam, torch.zeros_like(gam), torch.float32)
    dy = rnd(M, Cin, seed=3)
    dx, dxm = g.clone(), torch.empty(M, Cin, device=DEV)
    ops.layernorm_bwd(dy, x, st, gam, dx, True, torch.zeros_like(gam), torch.zeros_like(gam), dx_masked=dxm, mask=dd)
    assert torch.equal(dxm, dx * y)
    # gather layout: rows of the merged grid scatter back into [B, H, W, Cin] tokens
    gather = (B, H, W, Cin)
    g4 = rnd(4 * Cin, seed=6)
    _, st4 = ops.layernorm_fwd(x, g4, torch.zeros_like(g4), torch.float32, gather=gather)
    dy4 = rnd(M // 4, 4 * Cin, seed=7)
    dx4, dxm4 = torch.empty(M, Cin, device=DEV), torch.empty(M, Cin, device=DEV)
    ops.layernorm_bwd(dy4, x, st4, g4, dx4, False, torch.zeros_like(g4), torch.zeros_like(g4), gather=gather, dx_masked=dxm4, mask=dd)
    assert torch.equal(dxm4, dx4 * y)


@pytest.mark.parametrize("B,H", [(40, 256), (43, 256), (600, 256), (16, 128)])
def test_gru_whole_sequence_kernels_match_per_step_path(ops, B, H):
    """focal_gru_seq_fwd / _bwd (one launch per layer, both directions, bf16 W_hh) against the per-step GEMM + gate kernels
    given the same bf16 operands.  H = 256, B = 40 / 43: two lanes per sample, 8 samples per workgroup (43: a ragged last workgroup, its
    spare lanes recompute the last sample); B = 600: 16 samples per workgroup (the grid of the 8-sample form would exceed 64), ragged."""
    T = 10
    f32c, bfc = ops.code(torch.float32), ops.code(torch.bfloat16)
    gd = ops.GRUDesc(B, T, H)
    gi = [rnd(B * T, 3 * H, seed=300 + d) for d in range(2)]
    whh = [rnd(3 * H, H, scale=H ** -0.5, seed=310 + d) for d in range(2)]
    bhh = [rnd(3 * H, scale=0.1, seed=320 + d) for d in range(2)]
    w16 = [w.bfloat16() for w in whh]
    d_hh = ops.linear_desc(bfc, B, 3 * H, H, f32c, f32c)
    ref = dict(out=torch.zeros(B, T, 2 * H, device=DEV), hs=[], save=[])
    for di in range(2):
        hs = torch.zeros(T + 1, B, H, device=DEV)
        save = torch.empty(T, 4, B, H, device=DEV)
        gh = torch.empty(B, 3 * H, device=DEV)
        for s in range(T):
            t = s if di == 0 else T - 1 - s
            ops.linear_fwd(d_hh, hs[s], w16[di], bhh[di], None, gh)
            ops.gru_gate_fwd(gd, t, di * H, gi[di], gh, hs[s], hs[s + 1], ref["out"], save[s])
        ref["hs"].append(hs)
        ref["save"].append(save)
    out = torch.zeros(B, T, 2 * H, device=DEV)
    hs = [torch.zeros(T + 1, B, H, device=DEV) for _ in range(2)]
    save = [torch.empty(T, 4, B, H, device=DEV) for _ in range(2)]
    ops.gru_seq_fwd(gd, gi, w16, bhh, hs, save, out)
    assert rel_err(out, ref["out"]) < 2e-3
    for di in range(2):
        assert rel_err(hs[di], ref["hs"][di]) < 2e-3 and rel_err(save[di], ref["save"][di]) < 2e-3
    # backward from the SAME saved state: layer-0 style dout [B*T, 2H]
    dout = rnd(B * T, 2 * H, seed=330)
    ld_b, ld_t = T * 2 * H, 2 * H
    got = [(torch.empty(B * T, 3 * H, device=DEV), torch.empty(T, B, 3 * H, device=DEV)) for _ in range(2)]
    whh_t = [ops.permute_pack(w, 1, 3 * H, H, torch.bfloat16) for w in whh]
    ops.gru_seq_bwd(gd, dout, ld_b, ld_t, 1.0, whh_t, ref["hs"], ref["save"], [g[0] for g in got], [g[1] for g in got])
    for di in range(2):
        dgi = torch.empty(B * T, 3 * H, device=DEV)
        dgh = torch.empty(T, B, 3 * H, device=DEV)
        dhz = torch.empty(2, B, H, device=DEV)
        dh_rec = torch.empty(B, H, device=DEV)
        have = False
        for s in range(T - 1, -1, -1):
            t = s if di == 0 else T - 1 - s
            ops.gru_gate_bwd(gd, t, di * H, dout, ld_b, ld_t, 1.0, dh_rec if have else None, dhz[(s + 1) & 1] if have else None,
                             ref["save"][di][s], ref["hs"][di][s], dgi, dgh[s], dhz[s & 1])
            if s > 0:
                ops.linear_bwd_data(d_hh, dgh[s], w16[di], None, dh_rec)
                have = True
        assert rel_err(got[di][0], dgi) < 5e-3, di
        assert rel_err(got[di][1], dgh) < 5e-3, di
    # round 5: W_hh handed over in MFMA-fragment order (focal_pack_multi FOCAL_PACK_FRAG / _FRAG_T, focal_gru_desc.whh_frag): the same
    # fragments in the same registers -- bit-identical results
    frag = [torch.empty(3 * H * H, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    frag_t = [torch.empty(3 * H * H, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    ops.pack_multi([(whh[d], frag[d], 3 * H, H, 1, ops.PACK_FRAG) for d in range(2)] + [(whh[d], frag_t[d], 3 * H, H, 1, ops.PACK_FRAG_T) for d in range(2)],
                   torch.bfloat16)
    # the packed layout itself, against its definition
    m = w16[0]
    r, c = torch.meshgrid(torch.arange(3 * H, device=DEV), torch.arange(H, device=DEV), indexing="ij")
    idx = ((r // 16) * (H // 32) + c // 32) * 512 + ((c % 32) // 8 * 16 + r % 16) * 8 + c % 8
    want = torch.empty_like(frag[0])
    want[idx.reshape(-1)] = m.reshape(-1)
    assert torch.equal(frag[0], want)
    mt = w16[0].t().contiguous()
    r, c = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(3 * H, device=DEV), indexing="ij")
    idx = ((r // 16) * (3 * H // 32) + c // 32) * 512 + ((c % 32) // 8 * 16 + r % 16) * 8 + c % 8
    want[idx.reshape(-1)] = mt.reshape(-1)
    assert torch.equal(frag_t[0], want)
    gdf = ops.GRUDesc(B, T, H, 1)
    out_f = torch.zeros(B, T, 2 * H, device=DEV)
    hs_f = [torch.zeros(T + 1, B, H, device=DEV) for _ in range(2)]
    save_f = [torch.empty(T, 4, B, H, device=DEV) for _ in range(2)]
    ops.gru_seq_fwd(gdf, gi, frag, bhh, hs_f, save_f, out_f)
    assert torch.equal(out_f, out) and all(torch.equal(a, b) for a, b in zip(hs_f, hs)) and all(torch.equal(a, b) for a, b in zip(save_f, save))
    got_f = [(torch.empty(B * T, 3 * H, device=DEV), torch.empty(T, B, 3 * H, device=DEV)) for _ in range(2)]
    ops.gru_seq_bwd(gdf, dout, ld_b, ld_t, 1.0, frag_t, ref["hs"], ref["save"], [g[0] for g in got_f], [g[1] for g in got_f])
    for di in range(2):
        assert torch.equal(got_f[di][0], got[di][0]) and torch.equal(got_f[di][1], got[di][1])


@pytest.mark.parametrize("name", ["negation", "scaling", "horizontal_flip", "permutation", "phase_shift"])
def test_augmentation_folded_into_dft_matches_reference(ops, name):
    """focal_augment_fft_fwd against outputs of the reference augmenter classes (forced draws) followed / preceded by the
    reference FFT packing: tests/golden/augment_b2_seed77.npz."""
    import numpy as np
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_b2_seed77.npz"))
    kw = {"negation": dict(scale=-1.0), "scaling": dict(scale=float(fx["draw.scaling"])), "horizontal_flip": dict(flip=True),
          "permutation": dict(perm=[int(v) for v in fx["draw.permutation"]]), "phase_shift": dict(phase=float(fx["draw.phase_shift"]))}[name]
    for k in [k[3:] for k in fx.files if k.startswith("in.")]:
        x = torch.from_numpy(fx[f"in.{k}"]).to(DEV)
        ref = torch.from_numpy(fx[f"{name}.{k}"]).to(DEV)
        out = ops.fft_realpack(x, **kw)
        n = x.shape[-1]
        assert (out.double() - ref.double()).abs().max().item() < 2e-4 * math.sqrt(n) * max(1.0, abs(kw.get("scale", 1.0))), (name, k)
    # combinations compose: x' = permute(flip(scale * x)) before the transform, rotation after it
    x = rnd(2, 1, 10, 1600, seed=88)
    perm = [9, 8, 0, 1, 2, 3, 7, 6, 5, 4]
    got = ops.fft_realpack(x, scale=0.5, flip=True, perm=perm, phase=0.3)
    xk = torch.flip(0.5 * x.cpu(), dims=[2, 3])[:, :, torch.tensor(perm), :]
    f = torch.view_as_real(torch.fft.fft(xk.double(), dim=-1)).permute(0, 1, 4, 2, 3).reshape(2, 2, 10, 1600)
    c, s_ = math.cos(0.3), math.sin(0.3)
    ref = torch.stack([f[:, 0] * c - f[:, 1] * s_, f[:, 0] * s_ + f[:, 1] * c], 1)
    assert (got.cpu().double() - ref).abs().max().item() < 2e-4 * 40


def test_product_augmenter_random_views_follow_the_oracle(ops, cfg, monkeypatch):
    """data_augmenter.Augmenter.forward("random"): whatever it draws, the tensor it returns is the oracle's view for that draw."""
    from conftest import make_args
    from data_augmenter import Augmenter as A
    from oracle import augment as oa
    args = make_args(cfg, "SW_Transformer", torch.device(DEV), "bf16")
    aug = A.Augmenter(args)
    calls = []
    real = A.ops.fft_realpack_multi

    def spy(items):  # the augmenter hands all (location, modality) transforms of a view to ONE call
        calls.extend((it["x"], {k: v for k, v in it.items() if k not in ("x", "out")}) for it in items)
        return real(items)
    monkeypatch.setattr(A.ops, "fft_realpack_multi", spy)
    import random
    import numpy as np
    random.seed(12)  # the augmenter draws from `random`, numpy and torch: fix all three so the coverage check below is deterministic
    np.random.seed(12)
    torch.manual_seed(12)
    tx = {"shake": {"audio": torch.randn(2, 1, 10, 1600), "seismic": torch.randn(2, 1, 10, 20)}}
    seen = set()
    for _ in range(64):
        calls.clear()
        out = aug.forward("random", tx)
        assert list(out["shake"].keys()) == ["audio", "seismic"]
        for (x, kw), m in zip(calls, ("audio", "seismic")):
            seen.update(kw.keys())
            ref = x.cpu()
            if "scale" in kw:
                ref = oa.scaling(ref, kw["scale"])
            if kw.get("flip"):
                ref = oa.horizontal_flip(ref)
            if "perm" in kw:
                ref = oa.permutation(ref, kw["perm"])
            f = oa.fft_realpack(ref)
            if "phase" in kw:
                f = oa.phase_shift(f, kw["phase"])
            assert (out["shake"][m].cpu() - f).abs().max().item() < 2e-4 * math.sqrt(x.shape[-1]) * 2
    assert {"scale", "flip", "perm", "phase"} <= seen  # 64 seeded draws from a 7-entry pool at p = 0.5 each: all four kinds appear


# ---- round 5: the random views drawn on the device, inside the step (focal_view_draw / focal_warp_plan_multi / the transform's `plan`)
FOCAL_POOL = [("permutation", 0.5), ("negation", 0.5), ("time_warp", 0.5), ("horizontal_flip", 0.5), ("mag_warp", 0.5), ("scaling", 0.5),
              ("phase_shift", 0.5)]


def test_device_view_draws_have_the_reference_distributions(ops):
    """focal_view_draw over 6000 seed words, the shipped FOCAL pool: the augmenter of a view is uniform over the pool and shared by the
    view's slots (Augmenter.py:86), a slot's coin hits with p = 0.5 independently of the other slot (each augmenter class's random() <
    prob), the scale factors are N(1, 0.2) (ScalingAugmenter.py:35-36), the phase angles uniform on (-pi, pi) (PhaseShiftAugmenter.py:
    39-54), the interval orders uniform permutations (torch.randperm), the warp knots N(1, magnitude) with tsai's knot counts; the two
    views of a step and consecutive seeds are uncorrelated; the same seed gives the same plans."""
    import numpy as np
    from focal_amd import _lib
    pool = ops.view_pool(FOCAL_POOL, [10, 10])
    n_views, n_slots, N = 2, 2, 6000
    seed = ops.new_rng_state(1, DEV)
    plans = ops.new_view_plans(n_views, n_slots, DEV)
    recs = []
    for i in range(N):
        seed[0] = 7919 * i + 13
        ops.view_draw(pool, n_views, n_slots, seed, 0x56494557, plans)
        recs.append(ops.read_view_plans(plans))
    again = ops.read_view_plans(ops.view_draw(pool, n_views, n_slots, seed, 0x56494557, ops.new_view_plans(n_views, n_slots, DEV)))
    assert all(bytes(a) == bytes(b) for a, b in zip(again, recs[-1]))                       # a pure function of (seed, stream, view, slot)
    other = ops.read_view_plans(ops.view_draw(pool, n_views, n_slots, seed, 0x56494557 + 977, ops.new_view_plans(n_views, n_slots, DEV)))
    assert any(bytes(a) != bytes(b) for a, b in zip(other, recs[-1]))                       # another rank's stream draws differently
    k = np.array([[r[v * n_slots].pool_index for v in range(n_views)] for r in recs])       # [N, views]
    for r in recs:
        assert all(r[v * n_slots + s].pool_index == r[v * n_slots].pool_index for v in range(n_views) for s in range(n_slots))
    freq = np.bincount(k.ravel(), minlength=7) / k.size
    assert np.abs(freq - 1 / 7).max() < 0.012, freq                                         # (sigma = 0.0032 at 12 000 draws)
    assert abs(np.corrcoef(k[:, 0], k[:, 1])[0, 1]) < 0.04 and abs(np.corrcoef(k[:-1, 0], k[1:, 0])[0, 1]) < 0.04
    hit = np.array([[int(r[v * n_slots + s].kind != 0) for s in range(n_slots)] for r in recs for v in range(n_views)])
    assert np.abs(hit.mean(0) - 0.5).max() < 0.015 and abs(np.corrcoef(hit[:, 0], hit[:, 1])[0, 1]) < 0.03
    flat = [p for r in recs for p in r]
    kinds = {name: ops.VIEW_KINDS[name] for name, _ in FOCAL_POOL}
    for p in flat:   # the record applies exactly what was drawn, the identity otherwise
        ident = p.aug.scale == 1.0 and p.aug.flip == 0 and p.aug.use_perm == 0 and p.aug.phase_cos == 1.0 and p.aug.phase_sin == 0.0 and p.warp == 0
        assert (p.kind == 0) == ident or p.kind in (kinds["scaling"], kinds["phase_shift"])   # (a drawn factor / angle may be the identity by chance)
    sc = np.array([p.aug.scale for p in flat if p.kind == kinds["scaling"]])
    assert len(sc) > 1200 and abs(sc.mean() - 1.0) < 0.02 and abs(sc.std() - 0.2) < 0.015
    assert all(p.aug.scale == -1.0 for p in flat if p.kind == kinds["negation"]) and all(p.aug.flip == 1 for p in flat if p.kind == kinds["horizontal_flip"])
    ang = np.array([math.atan2(p.aug.phase_sin, p.aug.phase_cos) for p in flat if p.kind == kinds["phase_shift"]])
    assert len(ang) > 1200 and abs(ang.mean()) < 0.15 and abs(ang.std() - math.pi / math.sqrt(3)) < 0.08
    assert np.abs(np.histogram(ang, bins=8, range=(-math.pi, math.pi))[0] / len(ang) - 0.125).max() < 0.035
    perms = np.array([[p.aug.perm[i] for i in range(10)] for p in flat if p.kind == kinds["permutation"]])
    assert len(perms) > 1200 and all(sorted(row) == list(range(10)) for row in perms[:200])
    pos = np.stack([(perms == v).mean(0) for v in range(10)])                                # P(value v at position i) = 1 / 10
    assert np.abs(pos - 0.1).max() < 0.035, pos
    for name, nk, mag in (("mag_warp", 10, 0.05), ("time_warp", 16, 0.2)):
        kn = np.array([[p.knots[i] for i in range(nk)] for p in flat if p.kind == kinds[name]])
        assert len(kn) > 1200 and all(p.nknots == nk and p.warp == kinds[name] for p in flat if p.kind == kinds[name])
        assert abs(kn.mean() - 1.0) < 0.01 and abs(kn.std() - mag) < 0.04 * mag + 0.002
        assert abs(np.corrcoef(kn[:, 0], kn[:, 1])[0, 1]) < 0.08


@pytest.mark.parametrize("name", ["negation", "scaling", "horizontal_flip", "permutation", "phase_shift"])
def test_forced_device_plans_match_the_reference_augmenters(ops, name):
    """The transform reading its augmentation from a DEVICE plan record (what the captured step does) reproduces the reference augmenter
    classes' outputs for their forced draws: tests/golden/augment_b2_seed77.npz, the fixture of test_augmentation_folded_into_dft."""
    import numpy as np
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_b2_seed77.npz"))
    kw = {"negation": dict(scale=-1.0), "scaling": dict(scale=float(fx["draw.scaling"])), "horizontal_flip": dict(flip=True),
          "permutation": dict(perm=[int(v) for v in fx["draw.permutation"]]), "phase_shift": dict(phase=float(fx["draw.phase_shift"]))}[name]
    keys = [k[3:] for k in fx.files if k.startswith("in.")]
    plans = ops.new_view_plans(1, len(keys), DEV)
    items = []
    for i, k in enumerate(keys):
        ops.write_view_plan(plans, i, **kw)
        x = torch.from_numpy(fx[f"in.{k}"]).to(DEV)
        items.append(dict(x=x, plan=plans[i], x_warped=x))
    outs = ops.fft_realpack_multi(items)    # (the long rows on the MFMA kernel, the 20-sample rows in the shared small-row launch)
    for k, out in zip(keys, outs):
        ref = torch.from_numpy(fx[f"{name}.{k}"]).to(DEV)
        assert (out.double() - ref.double()).abs().max().item() < 2e-4 * math.sqrt(out.shape[-1]) * max(1.0, abs(kw.get("scale", 1.0))), (name, k)
        host = ops.fft_realpack(items[keys.index(k)]["x"], **kw)
        assert torch.equal(out, host), (name, k)   # the same kernels, the same values: only where they read them from differs


@pytest.mark.parametrize("I,S,C", [(10, 1600, 1), (10, 20, 2)])
def test_device_built_warp_tables_match_the_host_statement(ops, I, S, C):
    """focal_warp_plan_multi: the curve and the cumulative positions built ON THE DEVICE from a plan's knots equal focal_amd/warp.py's
    for the same knots (the host statement test_spline_warps_on_device_match_the_oracle pins against the oracle), and the warped rows --
    the pass forms the 24 weights of a position itself -- equal the oracle's; a plan without a warp leaves y alone."""
    import numpy as np
    from focal_amd import _lib, warp
    from oracle import augment
    x = rnd(8, C, I, S, seed=61)
    L = I * S
    plans = ops.new_view_plans(1, 3, DEV)
    kn_m = warp.draw_knots(4, 0.05, np.random.RandomState(2)).astype(np.float32)
    kn_t = warp.draw_knots(6, 0.2, np.random.RandomState(3)).astype(np.float32)
    ops.write_view_plan(plans, 0, warp=_lib.VIEW_MAG_WARP, knots=kn_m, kind=_lib.VIEW_MAG_WARP)
    ops.write_view_plan(plans, 1, warp=_lib.VIEW_TIME_WARP, knots=kn_t, kind=_lib.VIEW_TIME_WARP)
    ops.write_view_plan(plans, 2)
    tabs = [torch.zeros(2 * L, device=DEV) for _ in range(3)]
    ys = [torch.full_like(x, 7.0) for _ in range(3)]
    ops.warp_plan_multi([dict(x=x, plan=plans[i], tables=tabs[i], y=ys[i]) for i in range(3)])
    mult = warp.random_curve(L, kn_m.astype(np.float64), 4)
    assert np.abs(tabs[0][:L].cpu().numpy() - mult).max() < 2e-6
    pos = warp.warp_positions(L, kn_t.astype(np.float64), 6)
    pos_dev = tabs[1][:L].view(torch.int32).cpu().numpy().astype(np.float64) + tabs[1][L:].cpu().numpy().astype(np.float64)
    assert np.abs(pos_dev - pos).max() < 1e-6 * L, np.abs(pos_dev - pos).max()   # (fp64 on both sides; the fraction travels as fp32)
    assert pos_dev[0] == 0.0 and abs(pos_dev[-1] - (L - 1)) < 1e-9
    ref_m, ref_t = augment.mag_warp(x.cpu(), kn_m.astype(np.float64), 4), augment.time_warp(x.cpu(), kn_t.astype(np.float64), 6)
    assert (ys[0].cpu() - ref_m).abs().max().item() < 1e-5 * ref_m.abs().max().item()
    assert (ys[1].cpu() - ref_t).abs().max().item() < 5e-5 * ref_t.abs().max().item()
    assert torch.all(ys[2] == 7.0)                      # no warp drawn: nothing written
    # and the transform follows the plan's source choice: warped rows for plans 0 / 1, the plain rows for plan 2
    outs = ops.fft_realpack_multi([dict(x=x, plan=plans[i], x_warped=ys[i]) for i in range(3)])
    assert torch.equal(outs[0], ops.fft_realpack(ys[0])) and torch.equal(outs[1], ops.fft_realpack(ys[1])) and torch.equal(outs[2], ops.fft_realpack(x))


def test_product_augmenter_device_pair_follows_the_oracle(ops, cfg):
    """Augmenter.forward_random_pair (what train.py's step runs, captured): whatever the device drew, each view is the oracle's view for
    that draw; both views land in the halves of one tensor; over 40 seeds all seven augmenters of the FOCAL pool and the identity occur."""
    import numpy as np
    from conftest import make_args
    from data_augmenter import Augmenter as A
    from focal_amd import _lib, runtime
    from oracle import augment as oa
    args = make_args(cfg, "SW_Transformer", torch.device(DEV), "bf16")
    aug = A.Augmenter(args)
    assert aug.device_draws_supported()
    tx = {"shake": {"audio": torch.randn(2, 1, 10, 1600).to(DEV), "seismic": torch.randn(2, 1, 10, 20).to(DEV)}}
    seed = runtime.view_state(torch.device(DEV))   # (the draws' own state since round 6: one seed for all data-parallel ranks)
    saved = seed.clone()
    seen = set()
    try:
        for it in range(40):
            seed[0] = 1000 + 17 * it
            v = aug.forward_random_pair(tx)
            st = next(iter(aug._dev_states.values()))
            plans = ops.read_view_plans(st["plans"])
            for view in range(2):
                for i, m in enumerate(("audio", "seismic")):
                    p = plans[view * 2 + i]
                    seen.add(p.kind)
                    ref = tx["shake"][m].cpu()
                    if p.warp == _lib.VIEW_MAG_WARP:
                        ref = oa.mag_warp(ref, np.array([p.knots[j] for j in range(p.nknots)], np.float64), 4)
                    elif p.warp == _lib.VIEW_TIME_WARP:
                        ref = oa.time_warp(ref, np.array([p.knots[j] for j in range(p.nknots)], np.float64), 6)
                    if p.aug.scale != 1.0:
                        ref = oa.scaling(ref, p.aug.scale)
                    if p.aug.flip:
                        ref = oa.horizontal_flip(ref)
                    if p.aug.use_perm:
                        ref = oa.permutation(ref, [p.aug.perm[j] for j in range(10)])
                    f = oa.fft_realpack(ref)
                    if p.kind == _lib.VIEW_PHASE_SHIFT:
                        f = oa.phase_shift(f, math.atan2(p.aug.phase_sin, p.aug.phase_cos))
                    got = v[view]["shake"][m]
                    assert got.shape[0] == 2 and (got.cpu() - f).abs().max().item() < 2e-4 * math.sqrt(got.shape[-1]) * 2 + 1e-4 * f.abs().max().item(), (it, view, m, p.kind)
            both = st["both"][("shake", "audio")]
            assert v[0]["shake"]["audio"].data_ptr() == both.data_ptr() and v[1]["shake"]["audio"].data_ptr() == both[2:].data_ptr()
    finally:
        seed.copy_(saved)
    assert seen == set(range(8)), seen
    # a batch of another shape (the last batch of an epoch) gets a state of its own; the first one stays alive at its addresses: a captured
    # step graph keeps writing into it
    first = next(iter(aug._dev_states.values()))
    ptr = first["plans"].data_ptr()
    aug.forward_random_pair({"shake": {"audio": torch.randn(1, 1, 10, 1600).to(DEV), "seismic": torch.randn(1, 1, 10, 20).to(DEV)}})
    aug.forward_random_pair(tx)
    assert len(aug._dev_states) == 2 and next(iter(aug._dev_states.values()))["plans"].data_ptr() == ptr


def test_shared_view_draws_advance_their_own_state(ops):
    """focal_view_draw_shared (round 6): the draw reads its OWN 4-word state and moves it on (seed <- mix32(seed + golden), count += 1):
    two states started from one seed -- two data-parallel ranks -- draw identical plan sequences, consecutive calls differ, and call k
    equals focal_view_draw on the k-th word of the sequence."""
    pool = ops.view_pool(FOCAL_POOL, [10, 10])
    a, b = ops.new_rng_state(4242, DEV), ops.new_rng_state(4242, DEV)
    pa, pb = ops.new_view_plans(2, 2, DEV), ops.new_view_plans(2, 2, DEV)
    seq = []
    for k in range(6):
        word = a.clone()
        ops.view_draw_shared(pool, 2, 2, a, 0x56494557, pa)
        ops.view_draw_shared(pool, 2, 2, b, 0x56494557, pb)
        assert torch.equal(pa, pb) and torch.equal(a, b)
        assert int(a[1].item()) == k + 1 and int(a[0].item()) != int(word[0].item())
        plain = ops.view_draw(pool, 2, 2, word, 0x56494557, ops.new_view_plans(2, 2, DEV))
        assert torch.equal(plain, pa)
        seq.append(pa.cpu().clone())
    assert any(not torch.equal(seq[0], s) for s in seq[1:])
    with pytest.raises(ValueError):
        ops.write_view_plan(pa, 0, kind=6, warp=6)   # a warp without knots would be an out-of-bounds spline solve


def test_gpu_knn_matches_sklearn(ops):
    """train_utils.knn.GpuKNNClassifier (distances from the fp32 MFMA GEMM, top-5, majority vote, ties to the smallest label)
    against sklearn.neighbors.KNeighborsClassifier() -- the estimator the reference fits (train_utils/knn.py:38-40)."""
    from sklearn.neighbors import KNeighborsClassifier
    from train_utils.knn import GpuKNNClassifier
    g = torch.Generator().manual_seed(3)
    centers = torch.randn(7, 512, generator=g) * 0.12
    ytr = torch.randint(0, 7, (3000,), generator=g)
    xtr = centers[ytr] + torch.randn(3000, 512, generator=g)
    yq = torch.randint(0, 7, (1000,), generator=g)
    xq = centers[yq] + torch.randn(1000, 512, generator=g)
    ref = KNeighborsClassifier().fit(xtr.numpy(), ytr.numpy()).predict(xq.numpy())
    got = GpuKNNClassifier().fit(xtr.to(DEV), ytr.to(DEV)).predict(xq.to(DEV)).cpu().numpy()
    assert (got == ref).mean() >= 0.999, (got != ref).sum()   # (an exact distance tie at the 5th neighbour may order differently)
    assert 0.25 < (got == yq.numpy()).mean() < 0.999          # a real classification problem, neither trivial nor hopeless


@pytest.mark.parametrize("I,S,C", [(10, 1600, 1), (10, 20, 2)])
def test_spline_warps_on_device_match_the_oracle(ops, I, S, C):
    """TimeWarp / MagWarp (data_augmenter/{Time,Mag}WarpAugmenter.py -> tsai): focal_warp_fwd with the host-built tables vs the
    oracle's scipy restatement of tsai's algorithm with the same knot values."""
    import numpy as np
    from focal_amd import warp
    from oracle import augment
    x = rnd(8, C, I, S, seed=61)
    L = I * S
    kn = warp.draw_knots(4, 0.05, np.random.RandomState(2))
    mult = torch.from_numpy(warp.random_curve(L, kn, 4).astype(np.float32)).to(DEV)
    got = ops.mag_warp(x, mult)
    ref = augment.mag_warp(x.cpu(), kn, 4)
    assert (got.cpu() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    assert (got - x).abs().max().item() > 1e-3                 # not the identity
    kn = warp.draw_knots(6, 0.2, np.random.RandomState(3))
    k0, w = warp.time_warp_tables(warp.warp_positions(L, kn, 6))
    got = ops.time_warp(x, torch.from_numpy(k0).to(DEV), torch.from_numpy(w).to(DEV)).cpu().reshape(8, C, L)
    ref = augment.time_warp(x.cpu(), kn, 6).reshape(8, C, L)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 5e-5 * scale


def test_product_augmenter_draws_real_spline_warps(ops, cfg, monkeypatch):
    """`Augmenter.forward("random")` with the pool restricted to the two spline warps and the coin forced to heads: the view
    differs from the un-augmented spectrum (no identity stand-in any more) and equals the oracle's view for the knots it drew."""
    import copy
    import numpy as np
    import torch as T
    from conftest import make_args
    from data_augmenter import Augmenter as A
    from oracle import augment
    for name, order in (("mag_warp", 4), ("time_warp", 6)):
        c = copy.deepcopy(cfg)
        c["FOCAL"]["random_augmenters"] = {"time_augmenters": [name], "freq_augmenters": []}
        c[name]["prob"] = 1.0
        aug = A.Augmenter(make_args(c, "SW_Transformer", T.device(DEV)))
        drawn = []
        from focal_amd import warp
        real = warp.draw_knots

        def spy(order_, magnitude, rng=np.random):
            k = real(order_, magnitude, rng)
            drawn.append(k)
            return k
        monkeypatch.setattr(warp, "draw_knots", spy)
        x = {"shake": {"audio": rnd(4, 1, 10, 1600, seed=71).cpu(), "seismic": rnd(4, 1, 10, 20, seed=72).cpu()}}
        out = aug.forward("random", x)
        plain = aug.forward("no", x)
        assert len(drawn) == 2  # one curve per (location, modality)
        for i, mod in enumerate(c["modality_names"]):
            assert rel_err(out["shake"][mod], plain["shake"][mod]) > 1e-3
            ref = augment.augmented_view(x["shake"][mod], name, drawn[i])
            assert rel_err(out["shake"][mod].cpu(), ref) < 1e-4
        monkeypatch.setattr(warp, "draw_knots", real)
