"""Root-causing the finite-difference outlier of tests/test_swt_parity_gpu.py::test_dropout_on_backward_matches_forward_masks
(VERDICT r2 item 6): which evaluation after `p.add_(eps * d)` deviates, in which modality's output, by how many elements -- and does it
need the weight update at all, or only the allocation it makes (hypothesis: a kernel reads memory it has not written in this pass; in a
periodic allocation pattern that memory holds the previous pass's identical values, a freed temporary breaks the period).

  python tools/scratch/dbg_fd_outlier.py [rounds]          env: FOCAL_NO_STREAMS=1, DBG_SYNC=1 (synchronize after the update),
                                                                DBG_JUNK=1 (no weight update: allocate + free a junk tensor instead)
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import numpy as np
import torch
from conftest import make_args
from models.SW_Transformer import SW_Transformer
from oracle.config import load_config
from oracle.weights import fill_state_dict_, synthetic_freq_input

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = load_config()
args = make_args(cfg, "SW_Transformer", torch.device("cuda"), os.environ.get("DBG_CT", "fp32"))
net = SW_Transformer(args)
fill_state_dict_(net.state_dict())
net = net.to("cuda").train()
x = synthetic_freq_input(cfg, 4, seed=101)
x = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in x.items()}
params = dict(net.named_parameters())
names = ["freq_interval_layers.shake.audio.0.blocks.1.mlp.fc2.weight", "freq_interval_layers.shake.audio.1.blocks.0.attn.proj.weight",
         "freq_interval_layers.shake.seismic.2.blocks.3.mlp.fc1.weight", "freq_interval_layers.shake.audio.0.downsample.reduction.weight",
         "freq_interval_layers.shake.seismic.0.blocks.0.norm1.weight", "freq_interval_layers.shake.audio.2.blocks.2.attn.qkv.weight"]
sync = os.environ.get("DBG_SYNC") == "1"
junk = os.environ.get("DBG_JUNK") == "1"


def evaluate():
    net._fwd_calls = 0
    with torch.no_grad():
        out = net(x, class_head=False, proj_head=False)
    return {m: v.clone() for m, v in out.items()}


net.arena()
base = [evaluate() for _ in range(6)]
torch.cuda.synchronize()
mods = list(base[0])
spread = {m: max((base[i][m] - base[0][m]).abs().max().item() for i in range(6)) for m in mods}
print(f"unperturbed run-to-run max |diff| per modality: {spread}", flush=True)
events = 0
r = {m: torch.randn(4, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(i)) for i, m in enumerate(cfg["modality_names"])}
proj = lambda o: {m: (o[m] * r[m]).sum(dim=1).double().cpu().numpy() for m in mods}  # per (modality, sample) projected sums
psp = {m: max(np.abs(proj(base[i])[m] - proj(base[0])[m]).max() for i in range(6)) for m in mods}
print(f"unperturbed run-to-run max |diff| of the per-sample projected sums: {psp}", flush=True)
prev = proj(base[-1])
mode = os.environ.get("DBG_MODE", "clone")  # "item": exactly the test's value().item() (no clones, one scalar per evaluation)


def value_item():
    net._fwd_calls = 0
    with torch.no_grad():
        out = net(x, class_head=False, proj_head=False)
        return sum((out[m] * r[m]).sum() for m in out).item()


for rnd in range(rounds):
    for i, n in enumerate(names):
        p = params[n]
        d = torch.randn(p.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(100 + i))
        eps = 2e-3 / max(d.abs().max().item(), 1e-6)
        for sign in (+1, -2, +1):
            with torch.no_grad():
                if junk:
                    t = torch.full(p.shape, 1.0e3, device="cuda") * 1.0
                    del t
                else:
                    p.add_(sign * eps * d)
            if sync:
                torch.cuda.synchronize()
            if mode == "item":
                vs = [value_item() for _ in range(5)]
                med = float(np.median(vs))
                sp = max(abs(v - med) for v in vs)
                if sp > 2e-4:
                    events += 1
                    print(f"round {rnd} param {n} sign {sign}: values - median = {[f'{v - med:+.2e}' for v in vs]}", flush=True)
                continue
            outs = [proj(evaluate()) for _ in range(5)]
            ref = outs[-1]
            for k in range(4):
                for m in mods:
                    df = np.abs(outs[k][m] - ref[m])
                    if df.max() > 8 * max(psp[m], 1e-6):
                        events += 1
                        to_prev = np.abs(outs[k][m] - prev[m])
                        print(f"round {rnd} param {n} sign {sign}: evaluation {k} modality {m}: per-sample |diff to settled| {df}, "
                              f"|diff to the state BEFORE the update| {to_prev}  (update moved the sums by {np.abs(ref[m] - prev[m])})", flush=True)
            prev = ref
print(f"deviating evaluations: {events} (mode={mode} sync={sync} junk={junk} streams={'off' if os.environ.get('FOCAL_NO_STREAMS') == '1' else 'on'})")
