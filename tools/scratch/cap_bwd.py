import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_args
from input_utils.yaml_utils import load_yaml
from models.SW_Transformer import SW_Transformer
from focal_amd import runtime
mode = sys.argv[1]
cfg = load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml"))
args = make_args(cfg, "SW_Transformer", torch.device("cuda"), "bf16")
net = SW_Transformer(args).to("cuda").train()
B = 16
x = {"shake": {"audio": torch.randn(B, 2, 10, 1600, device="cuda"), "seismic": torch.randn(B, 2, 10, 20, device="cuda")}}
origin = torch.cuda.Stream()
def step():
    net.arena().zero_grad() if net._arena is not None else None
    out = net(x, class_head=False, proj_head=True)
    loss = sum(o.sum() for o in out.values())
    if mode == "direct":
        pass
    loss.backward()
    runtime.join_all(torch.device("cuda", 0))
with torch.cuda.stream(origin):
    step(); step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        step()
    g.replay(); torch.cuda.synchronize()
print("ok")
