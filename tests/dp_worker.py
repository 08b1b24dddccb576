"""Worker of tests/test_dp_parity_gpu.py: one rank of a world_size-2 data-parallel FOCAL step (fp32 mode, dropout off).

Both ranks share cuda:0 and talk over gloo (a 1-GPU box has no second device for RCCL; the collectives issued are the same
calls the RCCL job makes).  Rank r owns windows [r*B/2, (r+1)*B/2) of a seeded global batch.  Rank 0 then runs the CPU oracle
on the WHOLE batch as a single device would and checks: loss terms, every parameter gradient after the SUM all-reduce, and (for
DeepSense, sync-BN) the BatchNorm running statistics.  Exit code 0 = parity."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    model_name, B = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from conftest import make_args, no_dropout
    from input_utils.yaml_utils import load_yaml
    from focal_amd.distributed import all_reduce_gradients, gather_features
    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    from oracle.weights import fill_state_dict_, synthetic_freq_input
    cfg = no_dropout(load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml")))
    args = make_args(cfg, model_name, torch.device("cuda"), "fp32")
    if model_name == "DeepSense":
        from models.DeepSense import DeepSense as Net
        args.sync_bn = "nosync" not in sys.argv  # (manual negative control: per-rank statistics must NOT match)
    else:
        from models.SW_Transformer import SW_Transformer as Net
    net = Net(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
    g1, g2 = synthetic_freq_input(cfg, B, seed=101), synthetic_freq_input(cfg, B, seed=202)
    b = B // world
    mine = lambda d: {l: {m: v[rank * b:(rank + 1) * b].cuda() for m, v in mm.items()} for l, mm in d.items()}
    f1, f2 = focal(mine(g1), mine(g2), proj_head=True)
    f1, f2 = gather_features([f1, f2])
    net.arena().zero_grad()
    loss = loss_fn(f1, f2)
    loss.backward()
    from focal_amd import runtime
    runtime.join_all(torch.device("cuda", 0))
    all_reduce_gradients(net.arena())
    torch.cuda.synchronize()
    ok = True
    if rank == 0:
        from oracle.step import OracleTrainer
        state = {k: v.detach().cpu() for k, v in Net(args).state_dict().items()}
        fill_state_dict_(state)
        tr = OracleTrainer(model_name, cfg, state)
        terms, _, _, grads = tr.loss_and_grads(g1, g2)
        got = loss_fn.last_terms.cpu().tolist()
        for i, k in enumerate(("shared", "private", "orth", "rank", "total")):
            ref = float(terms[k])
            if abs(got[i] - ref) > 1e-3 * max(1.0, abs(ref)):
                print(f"LOSS MISMATCH {k}: {got[i]} vs {ref}")
                ok = False
        params = dict(net.named_parameters())
        worst = (0.0, None)
        for k, gref in grads.items():
            if gref is None:
                continue
            g = params[k].grad
            if g is None:
                print(f"MISSING GRAD {k}")
                ok = False
                continue
            scale = max(gref.abs().max().item(), 1e-6)
            if k.endswith("conv.bias") and scale < 1e-4:
                continue  # analytically zero in front of train-mode BatchNorm
            e = (g.detach().cpu() - gref).abs().max().item() / scale
            worst = max(worst, (e, k))
            if e > 2e-3:
                print(f"GRAD MISMATCH {k}: rel {e:.3e}")
                ok = False
        print(f"worst gradient error {worst[0]:.3e} ({worst[1]}); loss {got[4]:.6f} vs {float(terms['total']):.6f}")
        if model_name == "DeepSense":
            sd = net.state_dict()
            for k, v in tr.P.items():
                if k.endswith(("running_mean", "running_var")):
                    e = (sd[k].cpu() - v).abs().max().item() / max(v.abs().max().item(), 1e-6)
                    if e > 2e-4:
                        print(f"BN BUFFER MISMATCH {k}: {e:.3e}")
                        ok = False
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1 else 1)


if __name__ == "__main__":
    main()
