import sys, torch
dev = torch.device("cuda")
origin, mod, side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(1 << 20, device=dev); b = torch.zeros_like(a); c = torch.zeros_like(a)
flags = sys.argv[1]   # letters: s = initial join of side, m = initial join of mod, f = final join of side to origin, i = inner join side->mod
def worker():
    with torch.cuda.stream(mod):
        main = torch.cuda.current_stream()
        a.mul_(1.0001)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b.add_(a)
        if "i" in flags:
            main.wait_stream(side)
        c.copy_(a)
def body():
    cur = torch.cuda.current_stream()
    if "m" in flags: cur.wait_stream(mod)
    if "s" in flags: cur.wait_stream(side)
    a.add_(1.0)
    mod.wait_stream(cur)
    with torch.cuda.stream(mod):
        a.mul_(0.5)
    worker()
    cur.wait_stream(mod)
    if "f" in flags: cur.wait_stream(side)
    c.add_(1.0)
with torch.cuda.stream(origin):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        body()
    g.replay(); torch.cuda.synchronize()
print("ok", flags, float(c[0]))
