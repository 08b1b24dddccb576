"""Does any kernel of the step consume memory it has not written?  The allocator's cache is emptied and re-stocked with NaN-filled
blocks (one large block that later requests are carved from, plus a few thousand small ones for the small-block pool), then a
forward + backward runs: every torch.empty() of the pass then hands out NaN.  Outputs / gradients are compared with an unpoisoned
pass (a finite-difference outlier that appeared only inside the full pytest process, where the cache holds other
tests' data)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
from conftest import make_args
from oracle.config import load_config
from oracle.weights import fill_state_dict_, synthetic_freq_input

cfg = load_config()
POISON = float(os.environ.get("DBG_POISON", "nan"))


def poison(gb=24):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    small = [torch.full((s,), POISON, device="cuda") for s in (128, 1024, 8192, 65536, 200000) for _ in range(600)]
    big = torch.full((gb << 28,), POISON, device="cuda")  # gb GiB of fp32
    torch.cuda.synchronize()
    del small, big


def run(model, ct, B, do_poison):
    if model == "DeepSense":
        from models.DeepSense import DeepSense as Net
    else:
        from models.SW_Transformer import SW_Transformer as Net
    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    args = make_args(cfg, model, torch.device("cuda"), ct)
    torch.manual_seed(0)
    net = Net(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
    x1 = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in synthetic_freq_input(cfg, B, seed=101).items()}
    x2 = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in synthetic_freq_input(cfg, B, seed=202).items()}
    from focal_amd import runtime
    runtime.rng_state(torch.device("cuda"), seed=1234)
    net.arena().zero_grad()
    outs = []
    for it in range(3):
        if do_poison:
            poison()
        net._fwd_calls = 0
        runtime.rng_state(torch.device("cuda"), seed=1234)
        net.arena().zero_grad()
        f1, f2 = focal(x1, x2, proj_head=True)
        loss = loss_fn(f1, f2)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.item(), {m: f1[m].detach().clone() for m in f1}, net.arena().grad.clone(),
                     {k: v.clone() for k, v in net.named_buffers() if "running" in k}))
    return outs


for model, ct, B in (("SW_Transformer", "fp32", 8), ("SW_Transformer", "bf16", 8), ("SW_Transformer", "bf16", 24), ("SW_Transformer", "bf16", 64), ("DeepSense", "bf16", 8), ("DeepSense", "fp32", 8)):
    clean = run(model, ct, B, False)
    dirty = run(model, ct, B, True)
    l0, f0, g0, _ = clean[-1]
    noise_g = (clean[-1][2] - clean[-2][2]).abs().max().item()
    for it, (l, f, g, bufs) in enumerate(dirty):
        nan_f = {m: int(torch.isnan(v).sum().item()) for m, v in f.items()}
        nan_g = int(torch.isnan(g).sum().item())
        df = max((f[m] - f0[m]).abs().max().item() for m in f) if not any(nan_f.values()) else float("nan")
        dg = (g - g0).abs().max().item() if nan_g == 0 else float("nan")
        print(f"{model} {ct} B={B} poisoned pass {it}: loss {l:.6f} (clean {l0:.6f})  NaN in embeddings {nan_f}  NaN in gradients {nan_g}  "
              f"max |d embedding| {df:.2e}  max |d grad| {dg:.2e} (clean run-to-run {noise_g:.2e})", flush=True)
        if nan_g:
            idx = torch.isnan(g).nonzero().flatten()
            ar = net_ar = None
    print()
