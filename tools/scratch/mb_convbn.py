import sys, os, torch
sys.path.insert(0, os.getcwd())
from focal_amd import ops
DEV="cuda"; ct=torch.bfloat16
B,I,S,C,k=256,10,20,64,3
rows=B*I*S
x=torch.randn(rows,C,device=DEV).to(ct); w=torch.randn(C,C,1,k,device=DEV)*0.07; b=torch.randn(C,device=DEV)
d=ops.conv_desc(ops.code(ct),rows,S,C,C,k); w_fwd=ops.permute_pack(w,C,C,k,ct)
d_bn=ops.bn_desc(ops.code(ct),rows,C,I*S,0.0,None,0,momentum=0.1)
rm,rv=torch.zeros(C,device=DEV),torch.ones(C,device=DEV)
def timed(fn,n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/n*1e3
def two():
    ops.zero_pool_reset(DEV)
    z=ops.conv_fwd(d,x,w_fwd,b); ops.bn_stats(d_bn,z,rm,rv,True)
def one():
    ops.zero_pool_reset(DEV)
    ops.conv_fwd_bn(d,x,w_fwd,b,d_bn,rm,rv)
def conv_only():
    ops.zero_pool_reset(DEV)
    ops.conv_fwd(d,x,w_fwd,b)
print("conv only %.1f us, conv + bn_stats %.1f us, conv_fwd_bn %.1f us" % (timed(conv_only), timed(two), timed(one)))
