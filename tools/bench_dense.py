import sys, time, torch
sys.path.insert(0, ".")
from mop_amd.nn import EdgewiseMSA
def run(gate_mode, use_k3, B=256):
    torch.manual_seed(0)
    m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode=gate_mode, gate_rank=4, use_k3=use_k3).cuda().to(torch.bfloat16)
    x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn_like(x)
    for _ in range(2):
        m(x).backward(w)
    torch.cuda.synchronize()
    t = time.time()
    n = 3
    for _ in range(n):
        m(x).backward(w)
    torch.cuda.synchronize()
    print(gate_mode, "k3" if use_k3 else "", f"B={B}: {(time.time()-t)/n*1e3:.1f} ms/step, peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
run("lowrank", False)
run("dense", False)
run("dense", True)
