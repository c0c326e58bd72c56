import os, time, sys
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
import torch, bench
from oracle import edgewise_torch as oet
m = bench.build_layer_cpu()
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = torch.randn(16, 197, 384).requires_grad_(True); w = torch.randn(16, 197, 384)
    ts=[]
    for i in range(3):
        for t in list(p.values())+[x]: t.grad=None
        t0=time.perf_counter(); y=oet.edgewise_layer(x,p,6,5,0.5); (y*w).sum().backward(); ts.append(time.perf_counter()-t0)
    print("threads", nt, "per pass", [round(t,3) for t in ts], flush=True)
