import sys, threading, torch
prefork = sys.argv[1] == "1"
dev = torch.device("cuda")
origin, side = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(1 << 20, device=dev); b = torch.zeros_like(a); c = torch.zeros_like(a)
def worker():
    with torch.cuda.stream(origin):
        main = torch.cuda.current_stream()
        for i in range(8):
            a.mul_(1.0001)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b.add_(a)
        main.wait_stream(side)
        c.copy_(b)
def body():
    if prefork:
        side.wait_stream(torch.cuda.current_stream())
    t = threading.Thread(target=worker); t.start(); t.join()
    torch.cuda.current_stream().wait_stream(side)
with torch.cuda.stream(origin):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        body()
    g.replay(); torch.cuda.synchronize()
print("ok prefork", prefork, float(c[0]))
