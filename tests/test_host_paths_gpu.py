"""Host-side plumbing added around the HIP kernels: the per-step zero pool, the two-views-as-one-batch fast paths and their
fallbacks.  Each fast path must give the same numbers as the plain path it replaces."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))


def test_zero_pool_hands_out_zeroed_disjoint_slices_and_falls_back_when_exhausted():
    from focal_amd import ops
    dev = torch.device("cuda", 0)
    ops.zero_pool_reset(dev)
    a = ops.pool_zeros(100, dev)
    b = ops.pool_zeros(7, dev)
    assert a is not None and b is not None and a.numel() == 100 and b.numel() == 7
    assert a.data_ptr() % 256 == 0 and b.data_ptr() % 256 == 0 and b.data_ptr() >= a.data_ptr() + 400
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
    a.fill_(3.0)
    b.fill_(5.0)
    ops.zero_pool_reset(dev)  # one launch re-zeroes everything handed out since the last reset
    a2 = ops.pool_zeros(100, dev)
    assert a2.data_ptr() == a.data_ptr() and float(a2.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
    # exhaustion: callers get None and zero-fill themselves; ops.zeros hides the difference
    big = ops.pool_zeros(ops._ZERO_POOL_FLOATS, dev)
    assert big is None
    z = ops.zeros((ops._ZERO_POOL_FLOATS // 4096 + 1, 4096), dev)
    assert z.shape[1] == 4096 and float(z.abs().sum()) == 0.0
    ops.zero_pool_reset(dev)


def test_batchnorm_with_pool_scratch_equals_private_scratch():
    """focal_bn_stats / focal_bn_act_bwd with FOCAL_BN_SCRATCH_ZEROED (pool slice) vs their own memset."""
    from focal_amd import ops
    dev = torch.device("cuda", 0)
    rows, C = 5120, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    z = torch.randn(rows, C, generator=g).to(dev)
    gy = torch.randn(rows, C, generator=g).to(dev)
    gamma, beta = torch.rand(C, generator=g).to(dev) + 0.5, torch.randn(C, generator=g).to(dev)

    def run(use_pool):
        ops.zero_pool_reset(dev)
        if not use_pool:  # exhaust the pool so that both calls fall back to a private, self-zeroed scratch
            assert ops.pool_zeros(ops._ZERO_POOL_FLOATS - 64, dev) is not None
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        d = ops.bn_desc(ops.code(torch.float32), rows, C, 256, 0.0, None, 0)
        mr = ops.bn_stats(d, z, rm, rv, True)
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dz = ops.bn_act_bwd(d, z, gy, mr, gamma, beta, dgam, dbet, torch.float32)
        return mr.clone(), rm.clone(), rv.clone(), dz.clone(), dgam.clone(), dbet.clone()

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-4)  # (channel sums are fp32 atomics in arrival order)
    ops.zero_pool_reset(dev)


def test_views_as_one_batch_and_split_halves_fast_paths_match_the_plain_ones():
    from models.FOCALModules import _SplitHalves, _as_one_batch
    dev = "cuda"
    base = torch.randn(8, 3, 5, device=dev)
    x1, x2 = base[:4], base[4:]
    assert _as_one_batch(x1, x2) is base                       # adjacent halves of one tensor: no copy
    y1, y2 = x1.clone(), x2.clone()
    both = _as_one_batch(y1, y2)                               # unrelated tensors: concatenated
    assert both.data_ptr() not in (y1.data_ptr(), y2.data_ptr()) and torch.equal(both, base)
    assert _as_one_batch(base[:4], base[3:7]).shape[0] == 8 and _as_one_batch(base[:4], base[3:7]).data_ptr() != base.data_ptr()
    # split: forward halves; backward = one [2B, E] gradient, a view when the incoming gradients are adjacent
    f = torch.randn(6, 16, device=dev, requires_grad=True)
    h1, h2 = _SplitHalves.apply(f)
    flat = torch.randn(6 * 16, device=dev)
    g1, g2 = flat[:48].view(3, 16), flat[48:].view(3, 16)
    (gf,) = torch.autograd.grad([h1, h2], [f], [g1, g2])
    assert gf.data_ptr() == flat.data_ptr() and torch.equal(gf, flat.view(6, 16))
    f2 = f.detach().clone().requires_grad_(True)
    k1, k2 = _SplitHalves.apply(f2)
    (gf2,) = torch.autograd.grad([k1, k2], [f2], [g1.clone(), g2.clone()])  # separate allocations: concatenated
    assert torch.equal(gf2, flat.view(6, 16))
    ref = f.detach().clone().requires_grad_(True)
    (gr,) = torch.autograd.grad([ref[:3], ref[3:]], [ref], [g1, g2])
    assert torch.equal(gr, gf)


def test_dft_writes_views_into_one_batch_tensor():
    from focal_amd import ops
    x = torch.randn(4, 2, 10, 1600, device="cuda")
    both = torch.empty(8, 4, 10, 1600, device="cuda")
    a = ops.fft_realpack(x, out=both[:4])
    b = ops.fft_realpack(x, scale=-1.1, out=both[4:])
    assert a.data_ptr() == both.data_ptr() and b.data_ptr() == both[4:].data_ptr()
    assert torch.allclose(a, ops.fft_realpack(x)) and torch.allclose(b, ops.fft_realpack(x, scale=-1.1))
    with pytest.raises(ValueError):
        ops.fft_realpack(x, out=both[:3])


def test_augmenter_writes_back_to_back_views_into_one_batch_tensor():
    """Two consecutive `forward("random", same windows)` calls (the pretraining loop's two views) land in the two halves of one
    tensor, so FOCAL.forward can run them as one batch without a copy; any other call pattern simply gets fresh tensors."""
    import yaml
    from conftest import make_args
    from data_augmenter import Augmenter as A
    from models.FOCALModules import _as_one_batch
    with open(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml")) as f:
        cfg = yaml.safe_load(f)
    args = make_args(cfg, "SW_Transformer", torch.device("cuda"), "bf16")
    aug = A.Augmenter(args)
    loc = cfg["location_names"][0]
    x = {loc: {m: torch.randn(4, cfg["loc_mod_in_time_channels"][loc][m], cfg["num_segments"], cfg["loc_mod_spectrum_len"][loc][m],
                              device="cuda") for m in cfg["modality_names"]}}
    v1 = aug.forward("random", x)
    v2 = aug.forward("random", x)
    for m in cfg["modality_names"]:
        a, b = v1[loc][m], v2[loc][m]
        assert a._base is not None and a._base is b._base and b.data_ptr() == a.data_ptr() + a.numel() * 4
        assert _as_one_batch(a, b) is a._base
    other = {loc: {m: t.clone() for m, t in x[loc].items()}}
    v3 = aug.forward("random", other)   # different windows: starts a new pair
    v4 = aug.forward("random", x)       # not the partner of v3
    for m in cfg["modality_names"]:
        assert v3[loc][m]._base is not v4[loc][m]._base
        assert _as_one_batch(v3[loc][m], v4[loc][m]).shape[0] == 8  # concatenated


def test_static_views_pair_host_sourced_batches():
    """`static_views` (the graph-replaying loop): a loader that sources its batches on the HOST makes every `forward("random")`
    call move them to a NEW device tensor, so the two draws of a step cannot be paired by the input's address (the second
    view then overwrote the first one in the pooled tensor).  The halves are handed out in turn, and the step's tensor keeps its
    address from step to step."""
    import yaml
    from conftest import make_args
    from data_augmenter import Augmenter as A
    with open(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml")) as f:
        cfg = yaml.safe_load(f)
    args = make_args(cfg, "SW_Transformer", torch.device("cuda"), "bf16")
    aug = A.Augmenter(args)
    aug.static_views = True
    loc = cfg["location_names"][0]
    host = {loc: {m: torch.randn(4, cfg["loc_mod_in_time_channels"][loc][m], cfg["num_segments"], cfg["loc_mod_spectrum_len"][loc][m])
                  for m in cfg["modality_names"]}}
    bases = None
    for step in range(3):
        junk = torch.empty(1 << 20, device="cuda")  # perturbs the allocator between the two moves, as a warp's temporaries do
        v1 = aug.forward("random", host)
        del junk
        v2 = aug.forward("random", host)
        for m in cfg["modality_names"]:
            a, b = v1[loc][m], v2[loc][m]
            assert a._base is b._base and a.data_ptr() == a._base.data_ptr() and b.data_ptr() == a.data_ptr() + a.numel() * 4, (step, m)
        now = {m: v1[loc][m]._base.data_ptr() for m in cfg["modality_names"]}
        assert bases is None or bases == now
        bases = now
    # a lone draw (an exception between the two draws, a caller that wants one view) must not slip the pairing for good:
    # begin_step() -- what the training loop calls before a step's first draw -- puts view 1 back into the first half
    aug.forward("random", host)
    aug.begin_step()
    v1, v2 = aug.forward("random", host), aug.forward("random", host)
    for m in cfg["modality_names"]:
        a, b = v1[loc][m], v2[loc][m]
        assert a.data_ptr() == bases[m] and b.data_ptr() == bases[m] + a.numel() * 4, m


def test_launch_trace_records_kernels_with_their_own_durations():
    """focal_trace_* (include/focal_hip.h): bench.py's in-step roofline.  Between begin and end every launch of the library is recorded with
    the launched kernel's symbol and the dispatch's own duration; both modes see the same launches; a launch onto a stream that is being
    captured into a hipGraph is not recorded; durations are positive and a 75 MB AdamW-sized pass takes longer than a tiny one."""
    import ctypes
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_names import short_kernel_name
    from focal_amd import _lib, ops
    lib = _lib.load()
    big_x, small_x = torch.randn(1 << 22, 64, device="cuda"), torch.randn(256, 64, device="cuda")
    gamma, beta = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")

    def run():
        ops.layernorm_fwd(big_x, gamma, beta, torch.bfloat16)
        ops.layernorm_fwd(small_x, gamma, beta, torch.bfloat16)
    run()
    torch.cuda.synchronize()
    seen = {}
    for mode in (_lib.TRACE_DISPATCH, _lib.TRACE_EVENTS):
        _lib.check(lib.focal_trace_begin(64, mode))
        run()
        torch.cuda.synchronize()
        lib.focal_trace_end()
        n = lib.focal_trace_count()
        assert n == 2, n
        recs = (_lib.TraceRecord * n)()
        _lib.check(lib.focal_trace_read(0, n, recs))
        names = [short_kernel_name(r.kernel.decode()) for r in recs]
        assert all(nm.startswith("ln_fwd_kernel<bf16") for nm in names), names
        assert recs[0].us > 5 * recs[1].us > 0, (recs[0].us, recs[1].us)      # 1.6 GB moved against 100 KB
        assert recs[0].grid[0] >= recs[1].grid[0] and recs[0].block[0] == 256
        seen[mode] = [r.us for r in recs]
    # the big launch agrees between the two modes within the markers' own cost; and with the bytes it moves
    assert abs(seen[_lib.TRACE_DISPATCH][0] - seen[_lib.TRACE_EVENTS][0]) < 0.25 * seen[_lib.TRACE_DISPATCH][0]
    gbps = big_x.numel() * 6 / (seen[_lib.TRACE_DISPATCH][0] * 1e-6) / 1e9
    assert 1000 < gbps < 8000, gbps
    # nothing is recorded from a stream under capture, and an ended trace records nothing
    _lib.check(lib.focal_trace_begin(64, _lib.TRACE_DISPATCH))
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            run()
    assert lib.focal_trace_count() == 0
    lib.focal_trace_end()
    run()
    assert lib.focal_trace_count() == 0
    assert lib.focal_trace_begin(0, 1) != 0 and b"capacity" in lib.focal_last_error()
