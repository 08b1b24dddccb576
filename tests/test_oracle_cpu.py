"""CPU suite (no GPU): the oracle against the committed reference fixtures, host-side logic, and the C ABI surface
(library loads, exports every symbol include/focal_hip.h declares; no compute calls)."""
import json
import math
import os
import re

import numpy as np
import pytest
import torch

from conftest import make_args, no_dropout

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _state(model, cfg):
    from oracle import weights as ow
    spec = ow.swt_state_spec(cfg) if model == "SW_Transformer" else ow.deepsense_state_spec(cfg)
    st = {}
    for k, shp in spec.items():
        if k.endswith(("relative_position_index", "num_batches_tracked")):
            st[k] = torch.zeros(shp, dtype=torch.long)
        elif k.endswith("attn_mask"):
            st[k] = torch.zeros(shp)
        else:
            st[k] = ow.seeded_values(k, shp)
    return st


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_oracle_state_spec_matches_reference_manifest(cfg, model):
    from oracle import weights as ow
    man = json.load(open(os.path.join(GOLD, f"manifest_{model}.json")))
    spec = ow.swt_state_spec(cfg) if model == "SW_Transformer" else ow.deepsense_state_spec(cfg)
    assert [m[0] for m in man] == list(spec.keys())
    for k, shp, _ in man:
        assert tuple(shp) == tuple(spec[k]), k


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_oracle_eval_embeddings_match_reference(cfg, model):
    from oracle import weights as ow
    from oracle.deepsense import deepsense_forward
    from oracle.swt import swt_forward
    fx = np.load(os.path.join(GOLD, f"{model}_b8.npz"))
    st = _state(model, cfg)
    x1 = ow.synthetic_freq_input(cfg, 8, seed=101)
    taps = {}
    with torch.no_grad():
        if model == "SW_Transformer":
            emb = swt_forward(st, cfg, x1, proj_head=True, taps=taps)
        else:
            emb = deepsense_forward(st, cfg, x1, proj_head=True, train=False, taps=taps)
    for m in emb:
        ref = torch.from_numpy(fx[f"eval.emb.{m}"])
        assert (emb[m] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), m
    for k, v in taps.items():
        n = fx[f"eval.tapnorm.{k}"]
        assert abs(v.double().norm().item() - n[0]) < 1e-4 * max(n[0], 1e-6), k


def test_oracle_deepsense_eval_with_settled_statistics(cfg):
    """The oracle on the reference's own settled running statistics (tests/golden/gen_golden_deepsense_settled.py)."""
    from oracle import weights as ow
    from oracle.deepsense import deepsense_forward
    from conftest import no_dropout
    fx = np.load(os.path.join(GOLD, "DeepSense_settled_b8.npz"))
    c = no_dropout(cfg)
    st = _state("DeepSense", c)
    for k in fx.files:
        if k.startswith("buffer."):
            st[k[len("buffer."):]] = torch.from_numpy(fx[k])
    x1 = ow.synthetic_freq_input(c, 8, seed=101)
    with torch.no_grad():
        emb = deepsense_forward(st, c, x1, proj_head=True, train=False)
        feat = deepsense_forward(st, c, x1, proj_head=False, train=False)
    for m in emb:
        for got, key in ((emb[m], f"eval.emb.{m}"), (feat[m], f"eval.feat.{m}")):
            ref = torch.from_numpy(fx[key])
            assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), key


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_oracle_train_step_matches_reference(cfg, model):
    """Loss terms, per-parameter gradient norms and (DeepSense) BatchNorm running statistics of one FOCAL step."""
    from oracle import weights as ow
    from oracle.step import OracleTrainer
    fx = np.load(os.path.join(GOLD, f"{model}_b8.npz"))
    tr = OracleTrainer(model, cfg, _state(model, cfg))
    x1, x2 = ow.synthetic_freq_input(cfg, 8, seed=101), ow.synthetic_freq_input(cfg, 8, seed=202)
    terms, f1, f2, grads = tr.loss_and_grads(x1, x2)
    assert abs(float(terms["total"]) - float(fx["train.loss.reference_total"])) < 1e-4 * abs(float(fx["train.loss.reference_total"]))
    for k in ("shared", "private", "orth", "rank"):
        assert abs(float(terms[k]) - float(fx[f"train.loss.{k}"])) < 1e-5 * max(1.0, abs(float(fx[f"train.loss.{k}"])))
    for n, ref in zip(fx["train.grad_names"], fx["train.grad_norms"]):
        got = grads[str(n)].double().norm().item()
        assert abs(got - ref) < 5e-4 * ref + 1e-5, (n, got, ref)
    for k in fx.files:
        if k.startswith("train.buf."):
            name = k[len("train.buf."):]
            assert (tr.P[name] - torch.from_numpy(fx[k])).abs().max().item() < 1e-5 * max(1.0, float(np.abs(fx[k]).max()))


@pytest.mark.parametrize("model", ["SW_Transformer"])
def test_oracle_adamw_trajectory_matches_reference(cfg, model):
    from oracle import weights as ow
    from oracle.step import OracleTrainer
    fx = np.load(os.path.join(GOLD, f"{model}_b8.npz"))
    tr = OracleTrainer(model, cfg, _state(model, cfg))
    x1, x2 = ow.synthetic_freq_input(cfg, 8, seed=101), ow.synthetic_freq_input(cfg, 8, seed=202)
    for ref in fx["adamw.loss_traj"]:
        t = tr.step(freq_pair=(x1, x2))
        assert abs(t["total"] - ref) < 2e-3 * abs(ref)


@pytest.mark.parametrize("name,model", [("swt_b32", "SW_Transformer"), ("ds_b32", "DeepSense"), ("swt_b256", "SW_Transformer"),
                                        ("swt_4mod_b32", "SW_Transformer")])
def test_oracle_loss_matches_reference(cfg, name, model):
    from oracle.loss import focal_loss_terms
    fx = np.load(os.path.join(GOLD, f"loss_{name}.npz"))
    B, seed, scale = int(fx["B"]), int(fx["seed"]), float(fx["scale"])
    mods = [str(m) for m in fx["mods"]]
    g = torch.Generator().manual_seed(seed)
    f1 = {m: (torch.randn(B, 256, generator=g) * scale).requires_grad_(True) for m in mods}
    f2 = {m: torch.randn(B, 256, generator=g) * scale for m in mods}
    f2 = {m: (0.5 * f2[m] + 0.5 * f1[m].detach()).requires_grad_(True) for m in mods}
    import copy
    lcfg = copy.deepcopy(cfg)
    lcfg["modality_names"] = mods
    terms = focal_loss_terms(f1, f2, lcfg, model)
    assert abs(float(terms["total"]) - float(fx["loss.reference_total"])) < 2e-5 * abs(float(fx["loss.reference_total"]))
    terms["total"].backward()
    for m in mods:
        ref = torch.from_numpy(fx[f"demb1.{m}"])
        assert ((f1[m].grad - ref).norm() / ref.norm()).item() < 1e-4


def test_fft_realpack_matches_reference(cfg):
    from oracle import weights as ow
    from oracle.step import fft_realpack
    fx = np.load(os.path.join(GOLD, "fft_b4_seed11.npz"))
    out = fft_realpack(ow.synthetic_time_input(cfg, 4, seed=11))
    for loc in out:
        for mod in out[loc]:
            assert torch.equal(out[loc][mod], torch.from_numpy(fx[f"{loc}.{mod}"]))


# ---------------------------------------------------------------------------------------------- host logic
def test_product_state_dict_matches_reference_manifest(cfg):
    from models.SW_Transformer import SW_Transformer
    net = SW_Transformer(make_args(cfg, "SW_Transformer", torch.device("cpu")))
    man = json.load(open(os.path.join(GOLD, "manifest_SW_Transformer.json")))
    sd = net.state_dict()
    assert [m[0] for m in man] == list(sd.keys())
    for k, shp, dt in man:
        assert list(sd[k].shape) == shp and str(sd[k].dtype).replace("torch.", "") == dt, k
    # buffers derived from geometry agree with the oracle's restatement of the reference formulas
    from oracle.swt import relative_position_index, shifted_window_mask
    assert torch.equal(sd["freq_interval_layers.shake.audio.0.blocks.1.attn.relative_position_index"], relative_position_index(3, 3))
    assert torch.equal(sd["freq_interval_layers.shake.audio.0.blocks.1.attn_mask"], shifted_window_mask(12, 48, 3, 3, 1, 1))
    assert "freq_interval_layers.shake.audio.2.blocks.1.attn_mask" not in sd  # H = 3 <= window: no shift (the quirk)


def test_product_fails_loudly_without_gpu(cfg):
    """No CPU fallback anywhere on the product path."""
    from focal_amd import _lib
    from models.SW_Transformer import SW_Transformer
    net = SW_Transformer(make_args(cfg, "SW_Transformer", torch.device("cpu")))
    x = {"shake": {"audio": torch.zeros(4, 2, 10, 1600), "seismic": torch.zeros(4, 2, 10, 20)}}
    with pytest.raises(_lib.FocalHipError):
        net(x, class_head=False, proj_head=True)
    with pytest.raises(NotImplementedError):
        net(x, class_head=True)


def test_padded_size_and_schedules(cfg):
    from input_utils.padding_utils import get_padded_size
    from train_utils.lr_scheduler import CosineLRScheduler, StepLRScheduler
    assert get_padded_size((10, 1600), [3, 3], [1, 40], 3) == [12, 1920]
    assert get_padded_size((10, 20), [3, 3], [1, 1], 3) == [12, 24]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    s = CosineLRScheduler(opt, t_initial=6000, lr_min=1e-7)
    s.step(0)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-3)
    s.step(3000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-7 + 0.5 * (1e-3 - 1e-7) * (1 + math.cos(math.pi / 2)))
    s.step(7000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-7)
    from oracle.step import cosine_lr
    assert cosine_lr(1234, 1e-3, 1e-7, 6000) == pytest.approx(s.value(1e-3, 1234))
    st = StepLRScheduler(opt, decay_t=300, decay_rate=0.2)
    assert st.value(1e-4, 650) == pytest.approx(1e-4 * 0.2 ** 2)


def test_registry_and_errors(cfg):
    from train_utils.model_selection import init_backbone_model
    a = make_args(cfg, "NoSuchModel", torch.device("cpu"))
    with pytest.raises(Exception, match="Invalid model provided"):
        init_backbone_model(a)
    from input_utils.multi_modal_dataloader import SyntheticSequenceLoader
    with pytest.raises(ValueError):
        SyntheticSequenceLoader(make_args(cfg, "SW_Transformer", torch.device("cpu")), batch_size=30)
    dl = SyntheticSequenceLoader(make_args(cfg, "SW_Transformer", torch.device("cpu")), batch_size=8, num_batches=2)
    batches = list(dl)
    assert len(batches) == 2 and batches[0][0]["shake"]["audio"].shape == (8, 1, 10, 1600)


# ---------------------------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    from focal_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "focal_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(focal_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.load()  # raises if the .so is missing or a symbol does not resolve
    for name in declared:
        assert hasattr(lib, name)
    assert lib.focal_abi_version() == _lib.ABI_VERSION
    import ctypes as C
    assert C.sizeof(_lib.DropDesc) == 32 and C.sizeof(_lib.LinearDesc) == 9 * 4 + 4 + 32 + 8


def test_host_side_launch_plans_of_the_abi():
    """The planning entry points that touch no device: which kernel / how many workgroups a weight gradient launches, and the
    exchange chunk of the row-sharded loss head (with its argument checks)."""
    import ctypes as C
    from focal_amd import _lib
    lib = _lib.load()
    bf, f32 = _lib.FOCAL_BF16, _lib.FOCAL_F32

    def desc(M, N, K, dt, xdt, ydt):
        d = _lib.LinearDesc()
        d.dtype, d.M, d.N, d.K, d.x_dtype, d.y_dtype = dt, M, N, K, xdt, ydt
        return d
    d = desc(18432, 1024, 256, bf, bf, bf)           # whole 64-tiles in bf16: the LDS-DMA ring kernel, 64 tiles x 4 workgroups of 2 token slices
    assert lib.focal_linear_bwd_weight_kernel(C.byref(d)) == 2 and lib.focal_linear_bwd_weight_workgroups(C.byref(d)) == 256
    d.dw_workgroups = 192                            # the hint of a caller with several passes side by side: 64 tiles x 2
    assert lib.focal_linear_bwd_weight_workgroups(C.byref(d)) == 128
    d.dw_workgroups = 0
    assert lib.focal_linear_bwd_weight_workgroups(C.byref(d)) == 256
    d = desc(1000, 64, 256, bf, bf, bf)              # ragged token count: the register-staged kernel
    assert lib.focal_linear_bwd_weight_kernel(C.byref(d)) == 1
    d = desc(18432, 1024, 256, bf, f32, f32)         # fp32 operands rounded in the loader: register-staged
    assert lib.focal_linear_bwd_weight_kernel(C.byref(d)) == 1
    d = desc(18432, 1024, 256, f32, f32, f32)
    assert lib.focal_linear_bwd_weight_kernel(C.byref(d)) == 1
    # round 5: several weight gradients with fp32 operands behind one problem table (DeepSense's GRU: W_hh [768, 256] and W_ih [768, 128 /
    # 512] over 5 120 rows).  The launch's workgroup target is shared by the problems; a token slice is never shorter than 256 rows.
    probs = (_lib.DwProblem * 4)()
    for i, (M, N, K) in enumerate([(5120, 768, 256), (5120, 768, 128), (5120, 768, 512), (5120, 768, 256)]):
        probs[i] = _lib.DwProblem(1, 1, 1, None, M, N, K, 0)   # (non-null placeholders: the plan touches no memory)
    tiles = [12 * 4, 12 * 2, 12 * 8, 12 * 4]
    wg = lib.focal_linear_bwd_weight_group_f32_workgroups(bf, 4, probs, 1024)       # 256 per problem -> splits 6, 11, 3, 6
    assert wg == sum(t * s for t, s in zip(tiles, (6, 11, 3, 6)))
    assert lib.focal_linear_bwd_weight_group_f32_workgroups(bf, 4, probs, 100000) == sum(t * 20 for t in tiles)   # 5 120 / 256 = 20 slices at most
    assert lib.focal_linear_bwd_weight_group_f32_workgroups(bf, 9, probs, 0) == 0 and b"1 .. 8 problems" in lib.focal_last_error()
    # BatchNorm statistic groups: the descriptor grew by one field (ABI 10), the GRU descriptor by one (ABI 11); ABI 12 added focal_view_draw_shared,
    # ABI 13 the sums-only convolution statistics (focal_bn_act_fwd_sums) and 20 problems per grouped weight-gradient launch
    assert C.sizeof(_lib.BNDesc) == 56 and C.sizeof(_lib.GRUDesc) == 16 and lib.focal_abi_version() == 13
    ld = _lib.LossDesc(2, 2048, 256, 4, 0.07, 1.0, 1.0, 1.0, 3.0, 5.0, 0)  # config 4's global batch: b = 512 subsequences
    need = lib.focal_loss_head_workspace(C.byref(ld))
    assert need > 0
    n8, n1 = lib.focal_loss_head_exchange_floats(C.byref(ld), 8), lib.focal_loss_head_exchange_floats(C.byref(ld), 1)
    # a rank's chunk: lse of its 2 x 64 rows in each of the 4 problems x 4 steps, 4 x 64 diagonal means, 5 partial terms (64-padded)
    assert n8 == ((16 * 128 + 4 * 64 + 5 + 63) // 64) * 64 and n1 >= 8 * (n8 - 64)
    assert lib.focal_loss_head_exchange_floats(C.byref(ld), 3) == 0     # 512 subsequences do not split over 3 ranks
    assert b"split over 3 ranks" in lib.focal_last_error()


def test_augment_oracle_matches_reference_fixture():
    """oracle/augment.py against outputs of the reference augmenter classes with forced draws
    (tests/golden/gen_golden_augment.py): bit-exact for the time-domain augmenters, 1e-5 for the phase shift."""
    import numpy as np
    import torch
    from oracle import augment as oa
    fx = np.load(os.path.join(GOLD, "augment_b2_seed77.npz"))
    draws = {"negation": None, "scaling": float(fx["draw.scaling"]), "horizontal_flip": None,
             "permutation": [int(v) for v in fx["draw.permutation"]], "phase_shift": float(fx["draw.phase_shift"])}
    keys = [k[3:] for k in fx.files if k.startswith("in.")]
    assert keys
    for name, draw in draws.items():
        for lm in keys:
            x = torch.from_numpy(fx[f"in.{lm}"])
            ref = torch.from_numpy(fx[f"{name}.{lm}"])
            got = oa.augmented_view(x, name, draw)
            if name == "phase_shift":
                assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5
            else:
                assert torch.equal(got, ref), (name, lm)


def test_mixup_oracle_reproduces_reference_fixture():
    """oracle/augment.py::mixup_batch_random / mixup_bbox against the outputs of the reference's Mixup class with forced draws
    (tests/golden/mixup_b6.npz, generated by tests/golden/gen_golden_supervised.py)."""
    import numpy as np
    from oracle.augment import mixup_batch_random, mixup_bbox
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "mixup_b6.npz"))
    g = torch.Generator().manual_seed(int(fx["seed"]))
    base = {"audio": torch.randn(6, 1, 10, 64, generator=g), "seismic": torch.randn(6, 1, 10, 20, generator=g)}
    for tag, cut in (("mixup", False), ("cutmix", True)):
        lam, perm = float(fx[f"{tag}.lam"]), [int(v) for v in fx[f"{tag}.perm"]]
        for m, x in base.items():
            box = None
            if cut:
                box = mixup_bbox(x.shape[2], x.shape[3], lam, *[int(v) for v in fx[f"{tag}.centre.{m}"]])
                assert list(box) == [int(v) for v in fx[f"{tag}.box.{m}"]]
            assert torch.equal(mixup_batch_random(x, perm, lam, box), torch.from_numpy(fx[f"{tag}.out.{m}"]))


def test_kernel_symbol_decoder():
    """tools/kernel_names.py: bench.py's launch trace and `rocprofv3 -M` both give mangled symbols; both sides go through this decoder
    (the image's demanglers print __bf16 as `bool _Accum` or give up on DF16b)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_names import short_kernel_name as f
    assert f("_Z13ln_bwd_kernelIDF16bLi1EEvPKT_PKfS4_S4_PfiS4_S4_iii6RowMapPS0_15focal_drop_desci") == "ln_bwd_kernel<bf16, 1>"
    assert f("_Z22focal_gemm_pipe_kernelIDF16bLi6ELb0ELi128ELi128ELi2ELi2ELi2EEv10GemmParams") == "focal_gemm_pipe_kernel<bf16, 6, false, 128, 128, 2, 2, 2>"
    assert f("_ZN17focal_mlp_kernels14mlp_bwd_kernelILb1ELb1EEEv12MlpBwdParams") == "mlp_bwd_kernel<true, true>"
    assert f("_Z12adamw_kernelPfPKfS_S_PDF16blS1_PjS3_i16focal_adamw_desc") == "adamw_kernel"
    assert f("_Z17focal_gemm_kernelIDF16bffDF16bLb1ELb1ELi0ELi0ELi5ELi64ELi64ELi2EEv10GemmParams") == "focal_gemm_kernel<bf16, float, float, bf16, true, true, 0, 0, 5, 64, 64, 2>"
    assert f("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"
    # already (mis-)demangled input, as hipKernelNameRefByPtr / rocprofv3 without -M print it
    assert f("void (anonymous namespace)::mlp_bwd_kernel<true>(MlpBwdParams)") == "mlp_bwd_kernel<true>"
    assert f("void ln_bwd_kernel<bool _Accum, 1>(bool _Accum const*, float)") == "ln_bwd_kernel<bf16, 1>"
