import os, sys, torch, numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-wrf-les-ppo-lstm_amd")]
from uavppo import ops
def timeit(fn, n=5, burst=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(burst): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)/burst)
    return float(np.median(ts))
dev="cuda:0"; NT=1<<20
for M in (128, 256, 512, 1024):
    dg=torch.randn(NT, M, device=dev); y=torch.rand(NT,256,device=dev)
    t=timeit(lambda: ops.gemm(dg,y,trans_a=True,split_fp16=True))
    tiles=M//128; S=max(1,(256+tiles-1)//tiles); S=(S+7)//8*8
    slabs=NT/S/32
    print(f"M={M}: {t:.3f} ms, tiles {tiles} S {S}, {t*1e3/slabs:.2f} us per slab, {(NT*M*4+NT*256*4*tiles)/t/1e9:.2f} TB/s incl. B re-reads", flush=True)
