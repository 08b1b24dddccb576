import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import make_args, no_dropout
from oracle.config import load_config
from oracle.weights import fill_state_dict_
from models.DeepSense import DeepSense
cfg = load_config()
for ct in ("bf16",):
    args = make_args(no_dropout(cfg), "DeepSense", torch.device("cuda"), ct)
    net = DeepSense(args); fill_state_dict_(net.state_dict()); net = net.to("cuda").train()
    g = torch.Generator().manual_seed(7)
    x8 = {"shake": {"audio": torch.randn(8, 2, 10, 1600, generator=g).cuda(), "seismic": torch.randn(8, 2, 10, 20, generator=g).cuda()}}
    r8 = {m: torch.randn(8, 256, generator=g).cuda() for m in cfg["modality_names"]}
    for reps in (1, 2, 4, 32):
        x = {"shake": {m: v.repeat(reps, 1, 1, 1) for m, v in x8["shake"].items()}}
        r = {m: v.repeat(reps, 1) for m, v in r8.items()}
        net.arena().zero_grad()
        out = net(x, class_head=False, proj_head=False)
        if reps == 1:
            base = {m: out[m].detach().clone() for m in out}
        for m in out:
            print(ct, "reps", reps, m, "fwd feat max diff rows 0-8:", (out[m][:8] - base[m]).abs().max().item(), "last block:", (out[m][-8:] - base[m]).abs().max().item(), "scale", base[m].abs().max().item())
        sum((out[m] * r[m][:, :out[m].shape[1]] if out[m].shape[1] <= 256 else (out[m][:, :256] * r[m])).sum() for m in out).backward()
        torch.cuda.synchronize()
        gr = net.arena().grad.clone() / reps
        if reps == 1:
            g0 = gr.clone()
        else:
            ar = net.arena(); worst = []
            for name, (off, n, shape) in ar.index.items():
                if name.endswith("conv.bias"): continue
                a, b = gr[off:off+n], g0[off:off+n]
                worst.append(((a-b).norm().item()/max(b.norm().item(),1e-12), name))
            worst.sort(reverse=True)
            print("   grad vs reps=1:", [(round(w,4), n.split('.',2)[-1][-45:]) for w, n in worst[:5]])
