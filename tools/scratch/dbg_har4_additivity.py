"""Which gradients break additivity over batch halves on HAR4 at B = 256 in bf16 (intermittent: 2e-4 in one run, 1.9e-2 in another)?"""
import argparse, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
from conftest import no_dropout
from models.SW_Transformer import SW_Transformer
from oracle.config import load_config
from oracle.weights import fill_state_dict_
ct = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = no_dropout(load_config(os.path.join(ROOT, "focal_amd", "src", "data", "HAR4.yaml")))
args = argparse.Namespace(model="SW_Transformer", dataset="HAR4", device=torch.device("cuda"), train_mode="contrastive", learn_framework="FOCAL",
                          stage="pretrain", task="activity_classification", tag=None, dataset_config=cfg, compute_dtype=ct)
net = SW_Transformer(args)
fill_state_dict_(net.state_dict())
net = net.to("cuda").train()
B = 256
g = torch.Generator().manual_seed(7)
loc = cfg["location_names"][0]
x = {loc: {m: torch.randn(B, cfg["loc_mod_in_freq_channels"][loc][m], cfg["num_segments"], cfg["loc_mod_spectrum_len"][loc][m], generator=g).cuda() for m in cfg["modality_names"]}}
sub = lambda lo, hi: {loc: {m: v[lo:hi] for m, v in x[loc].items()}}
r = {m: torch.randn(B, cfg["FOCAL"]["emb_dim"], generator=g).cuda() for m in cfg["modality_names"]}


def grads(lo, hi):
    net.arena().zero_grad()
    out = net(sub(lo, hi), class_head=False, proj_head=True)
    sum((out[m] * r[m][lo:hi]).sum() for m in out).backward()
    torch.cuda.synchronize()
    return net.arena().grad.clone()


ar = net.arena()
for rep in range(8):
    gfull = grads(0, B)
    gfull2 = grads(0, B)
    g1, g2 = grads(0, B // 2), grads(B // 2, B)
    gmax = gfull.abs().max().item()
    d = (gfull - (g1 + g2)).abs()
    dd = (gfull - gfull2).abs()
    worst = []
    for n, (o, k, shp) in ar.index.items():
        e = d[o:o + k].max().item() / gmax
        worst.append((e, dd[o:o + k].max().item() / gmax, n))
    worst.sort(reverse=True)
    print(f"rep {rep}: additivity err {d.max().item() / gmax:.2e}, full vs full again {dd.max().item() / gmax:.2e}; worst: " + "; ".join(f"{e:.1e} (rerun {e2:.1e}) {n[-60:]}" for e, e2, n in worst[:4]), flush=True)
