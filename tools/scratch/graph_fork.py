import sys, torch
n_fork = int(sys.argv[1]); join_each = sys.argv[2] == "1"
dev = torch.device("cuda")
origin, side = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(1 << 20, device=dev); b = torch.zeros_like(a); c = torch.zeros_like(a)
def body():
    main = torch.cuda.current_stream()
    for i in range(n_fork):
        a.mul_(1.0001)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b.add_(a)
        if join_each:
            main.wait_stream(side)
    main.wait_stream(side)
    c.copy_(b)
with torch.cuda.stream(origin):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        body()
    g.replay(); torch.cuda.synchronize()
print("ok", n_fork, join_each, float(c[0]))
