"""`train.py` as a data-parallel job (the reference's `-gpu` flag parses device lists, params/params_util.py:34-38):
two ranks launched exactly as a user would (`torchrun --nproc-per-node 2 train.py ...`), sharing the box's one GPU over gloo
(FOCAL_DIST_BACKEND / FOCAL_DIST_ONE_DEVICE; on a multi-GPU box the same command runs RCCL, see test_rccl_two_gpus), must end an
epoch with the weights the CPU oracle reaches by running the WHOLE global batches on a single device."""
import copy
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "focal_amd", "src")


def _deterministic_yaml(cfg, tmp_path):
    from conftest import no_dropout
    c = no_dropout(copy.deepcopy(cfg))
    c["FOCAL"]["random_augmenters"] = {"time_augmenters": ["no"], "freq_augmenters": ["no"]}
    for k in ("pretrain_index_file",):
        c[k] = "synthetic"
    p = tmp_path / "MOD_det.yaml"
    p.write_text(yaml.safe_dump(c))
    return c, str(p)


def _launch(nproc, extra, env_extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(SRC, "train.py")] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=SRC)


@pytest.mark.parametrize("model,nb", [("SW_Transformer", 2), ("DeepSense", 2), ("SW_Transformer", 6)])
def test_two_rank_train_py_matches_whole_batch_oracle(cfg, tmp_path, model, nb):
    """nb = 6: long enough for the ranks to capture the step (after two eager steps) and REPLAY it -- hipGraph segments with the
    collectives issued eagerly between them (focal_amd/graph_step.py) -- for the rest of the epoch; the weights must still be the
    whole-batch oracle's."""
    from oracle.step import OracleTrainer
    from oracle.weights import fill_state_dict_, synthetic_time_input  # noqa: F401
    from conftest import make_args
    c, ypath = _deterministic_yaml(cfg, tmp_path)
    args = make_args(c, model, torch.device("cpu"))
    if model == "DeepSense":
        from models.DeepSense import DeepSense as Net
    else:
        from models.SW_Transformer import SW_Transformer as Net
    state = {k: v.detach().clone() for k, v in Net(args).state_dict().items()}
    fill_state_dict_(state)
    init = tmp_path / "init.pt"
    torch.save(state, str(init))
    B = 16  # global batch 16 = 2 ranks x 8 windows, nb steps in the epoch
    extra = [f"-model={model}", "-dataset=MOD", "-learn_framework=FOCAL", f"-batch_size={B}", f"-synthetic_batches={nb}", "-epochs=1",
             "-compute_dtype=fp32", f"-config={ypath}", f"-init_weight={init}"] + (["-sync_bn"] if model == "DeepSense" else [])
    r = _launch(2, extra, {"FOCAL_DIST_BACKEND": "gloo", "FOCAL_DIST_ONE_DEVICE": "1"}, 29561 if model == "DeepSense" else 29562 + nb)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    import re
    m = re.search(r"(\d+) graph replays, (\d+) eager steps", log)
    assert m, log[-2000:]
    # two eager steps (the second one ends with the capture), then replays
    assert int(m.group(1)) == max(0, nb - 2) and int(m.group(2)) == min(nb, 2), (m.group(0), [ln for ln in log.splitlines() if "capture" in ln or "Error" in ln][-5:])
    assert log.count("Val loss:") == 1, "validation must run on rank 0 only"   # rank 1 logs at WARNING level and skips the branch
    got = torch.load(os.path.join(ROOT, "weights", f"MOD_{model}", f"MOD_{model}_pretrain_latest.pt"), map_location="cpu")
    # the oracle on the whole global batches: rank r's windows of batch k are seeded 1234 + k + 100003 r (SyntheticSequenceLoader)
    tr = OracleTrainer(model, c, state)
    for k in range(nb):
        parts = []
        for rank in range(2):
            g = torch.Generator().manual_seed(1234 + k + 100003 * rank)
            batch = {}
            for loc in c["location_names"]:
                batch[loc] = {}
                for mod in c["modality_names"]:
                    shape = (B // 2, c["loc_mod_in_time_channels"][loc][mod], c["num_segments"], c["loc_mod_spectrum_len"][loc][mod])
                    batch[loc][mod] = torch.randn(shape, generator=g)
            parts.append(batch)
        whole = {loc: {mod: torch.cat([p[loc][mod] for p in parts]) for mod in parts[0][loc]} for loc in parts[0]}
        tr.step(time_x=whole)
    # AdamW's first steps move every weight by ~lr whatever the gradient's size (m / sqrt(v) = +-1), so the update itself is the
    # yardstick, in L2 per tensor (an element whose gradient is at rounding level may flip its sign: 2 lr on that element).  A
    # wrong exchange (local instead of global negatives, a missing rank's gradient) changes the sign pattern wholesale.
    def update_error(a, ref):
        worst, num, den = (0.0, None), 0.0, 0.0
        for key in tr.train_keys:
            d2 = (ref[key] - state[key]).double().pow(2).sum().item()
            e2 = (a[key] - ref[key]).double().pow(2).sum().item()
            num, den = num + e2, den + d2
            if d2 > 0 and ref[key].numel() >= 1024:
                worst = max(worst, ((e2 / d2) ** 0.5, key))
        return (num / max(den, 1e-300)) ** 0.5, worst, den
    err, worst, den = update_error(got, tr.P)
    from conftest import record_observed
    record_observed(f"train_dp.{model}.nb{nb}.update_error_vs_oracle", err)
    if nb <= 2:
        assert den > 0 and err < 0.03, (err, worst)
        assert worst[0] < 0.15, worst
    else:
        # Six steps: rounding-level sign flips of the first steps compound (every later step starts from weights that differ by a
        # few lr in a few elements), so the oracle comparison is loose here and the sharp check is against the SAME job launched
        # eagerly (-no_graph): replayed segments + eager collectives must reproduce eager steps up to the summation order of atomics
        assert den > 0 and err < 0.25, (err, worst)
        # (the control also keeps the backward pass whole and the gradient all-reduce in one blocking call: the replayed job splits the
        # pass where the last stage's gradients are final and reduces the arena in two buckets, the first beside the rest of backward)
        r2 = _launch(2, extra + ["-no_graph"], {"FOCAL_DIST_BACKEND": "gloo", "FOCAL_DIST_ONE_DEVICE": "1", "FOCAL_NO_SPLIT_BACKWARD": "1"}, 29580)
        assert r2.returncode == 0, (r2.stdout + r2.stderr)[-4000:]
        assert "0 graph replays" in r2.stdout + r2.stderr
        eager = torch.load(os.path.join(ROOT, "weights", f"MOD_{model}", f"MOD_{model}_pretrain_latest.pt"), map_location="cpu")
        err_e, worst_e, _ = update_error(eager, tr.P)
        err_ge, worst_ge, _ = update_error(got, eager)
        record_observed(f"train_dp.{model}.nb{nb}.eager_update_error_vs_oracle", err_e)
        record_observed(f"train_dp.{model}.nb{nb}.graph_vs_eager_update_error", err_ge)
        assert err_ge < max(0.05, 0.75 * err), (err_ge, worst_ge, err, err_e)
    if model == "DeepSense":
        bad = []
        for key, v in tr.P.items():
            if key.endswith(("running_mean", "running_var")):
                e = (got[key] - v).abs().max().item() / max(1.0, v.abs().max().item())
                if e > 2e-3:  # (the second step's statistics see weights that differ by the first step's AdamW sign noise)
                    bad.append((key, e, (got[key] - state[key]).abs().max().item(), (v - state[key]).abs().max().item()))
        assert not bad, bad  # (key, error, how far the job moved the buffer, how far the oracle moved it)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_rccl_two_gpus(cfg, tmp_path):
    """On a multi-GPU box: the same launch over RCCL ("nccl" backend, one GPU per rank) -- process-group creation bound to the
    device before the first HIP call, broadcast, all-gather of the embeddings, all-reduce of the arena, barrier, teardown."""
    c, ypath = _deterministic_yaml(cfg, tmp_path)
    extra = ["-model=SW_Transformer", "-dataset=MOD", "-learn_framework=FOCAL", "-batch_size=32", "-synthetic_batches=2", "-epochs=1",
             f"-config={ypath}"]
    r = _launch(2, extra, {}, 29563)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    assert log.count("Val loss:") == 1
    # and the benchmark's N = 2 path (3 hipGraph segments around the two eager collectives)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29564", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "32",
           "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert '"n_gpus": 2' in r.stdout


@pytest.mark.parametrize("ct", ["fp32", "bf16"])
def test_split_backward_first_phase_is_final_on_its_spans(cfg, ct):
    """The bucketed all-reduce (focal_amd/graph_step.py) starts on the spans `final_after_first_phase()` names as soon as the FIRST phase
    of the split backward pass has run, while `backward_continue()` still writes the arena.  One device, no process group:
    (1) after phase 1 alone those spans already hold the gradients of a whole backward pass -- bit for bit in fp32, and in bf16 too
    (the grouped weight-gradient launches with read-add-write "exclusive" stores, the PatchMerging gradient folded into the last
    stage's group and the fused LayerNorm-backward epilogues all run under split_backward there); (2) phase 2 writes nothing into them;
    (3) after phase 2 every span equals the whole pass."""
    from conftest import make_args, no_dropout
    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    from models.SW_Transformer import SW_Transformer
    from oracle.weights import fill_state_dict_, synthetic_freq_input
    args = make_args(no_dropout(cfg), "SW_Transformer", torch.device("cuda"), ct)
    net = SW_Transformer(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
    dev = lambda d: {l: {m: v.cuda() for m, v in mm.items()} for l, mm in d.items()}
    x1, x2 = dev(synthetic_freq_input(cfg, 16, seed=401)), dev(synthetic_freq_input(cfg, 16, seed=402))
    ar = net.arena()

    def backward(split):
        ar.zero_grad()
        f1, f2 = focal(x1, x2, proj_head=True)
        loss = loss_fn(f1, f2)
        net.split_backward = split
        try:
            loss.backward()
        finally:
            net.split_backward = False
        torch.cuda.synchronize()

    backward(False)
    whole = ar.grad.clone()
    first = set(net.final_after_first_phase())
    rest = [n for n in ar.index if n not in first]
    assert rest and first
    first_spans, rest_spans = ar.spans(first), ar.spans(rest)
    assert sum(hi - lo for lo, hi in first_spans) > 4 * sum(hi - lo for lo, hi in rest_spans)   # most gradient BYTES travel in the first bucket
    in_first = torch.zeros_like(whole, dtype=torch.bool)
    for lo, hi in first_spans:
        in_first[lo:hi] = True
    for lo, hi in rest_spans:
        assert not in_first[lo:hi].any()          # the two buckets do not overlap
    # run-to-run noise of the fp32 atomics (two whole passes differ by this much): the yardstick for "equal"
    backward(False)
    noise = (ar.grad - whole).abs().max().item() / whole.abs().max().item()
    backward(True)
    assert len(net.pending_backward) == len(cfg["modality_names"]), "split_backward must park one second half per encoder"
    phase1 = ar.grad.clone()
    # (bf16: two passes are not bit-equal -- the order of the fp32 atomics moves a few activations to the neighbouring bf16 value and
    # gradients by up to ~1e-2 of their maximum, DESIGN 4 -- so "equal" is that noise there; the exact check is (2) below)
    tol = max(10 * noise, 1e-6 if ct == "fp32" else 3e-2)
    scale = whole.abs().max().item()
    assert (phase1[in_first] - whole[in_first]).abs().max().item() < tol * scale, "a first-bucket gradient is not final after phase 1"
    # (phase 1 may already write second-bucket gradients -- in bf16 the PatchMerging reduction in front of the last stage gets its weight
    # gradient from the last stage's grouped launch -- which is harmless: that bucket is reduced after phase 2; what must hold is above)
    early = [n for n in rest if (phase1[ar.index[n][0]:ar.index[n][0] + ar.index[n][1]] != 0).any()]
    assert all("downsample.reduction" in n for n in early), early
    net.backward_continue()
    torch.cuda.synchronize()
    both = ar.grad.clone()
    assert torch.equal(both[in_first], phase1[in_first]), "phase 2 wrote into the first bucket (it is on the wire by then)"
    assert (both - whole).abs().max().item() < tol * scale


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_rccl_two_gpus_bf16_split_backward_equals_blocking_all_reduce(cfg, tmp_path):
    """Two ranks over RCCL, bf16 operands: six steps with the captured, split backward pass and the two-bucket all-reduce (the first
    bucket in flight on RCCL's stream while the graph-replayed rest of backward writes the other spans of the SAME arena) against the
    same job with FOCAL_NO_SPLIT_BACKWARD=1 (whole backward, one blocking all-reduce): same weights to bf16 run-to-run noise (ADVICE r3:
    the gloo test on one device cannot overlap the collective with the replay).  Skipped on the 1-GPU boxes of this build."""
    c, ypath = _deterministic_yaml(cfg, tmp_path)
    outs = []
    for i, env_extra in enumerate(({}, {"FOCAL_NO_SPLIT_BACKWARD": "1"})):
        extra = ["-model=SW_Transformer", "-dataset=MOD", "-learn_framework=FOCAL", "-batch_size=32", "-synthetic_batches=6", "-epochs=1",
                 "-compute_dtype=bf16", f"-config={ypath}"]
        r = _launch(2, extra, env_extra, 29571 + i)
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        outs.append(torch.load(os.path.join(ROOT, "weights", "MOD_SW_Transformer", "MOD_SW_Transformer_pretrain_latest.pt"), map_location="cpu"))
    a, b = outs
    worst = max(((a[k].float() - b[k].float()).abs().max().item() / max(1e-6, b[k].float().abs().max().item())) for k in a if a[k].is_floating_point())
    assert worst < 2e-2, worst
