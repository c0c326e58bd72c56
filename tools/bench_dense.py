"""dev tool (GPU box): EdgewiseMSA layer at the bench shape (B=256, N=197, D=384, H=6, V=5, bf16) with the low-rank head, the low-rank
head + S lens bank (dilations 1, 2), the dense head and dense + k3: training step (fwd+bwd) and inference forward, on the route the
module picks and on the generic path (set_path('generic'))."""
import sys, time, torch
sys.path.insert(0, ".")
from mop_amd import ops
from mop_amd.nn import EdgewiseMSA


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


def run(gate_mode, use_k3, B=256, lens=False):
    torch.manual_seed(0)
    m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode=gate_mode, gate_rank=4, use_k3=use_k3, use_lens_bank=lens).cuda().to(torch.bfloat16)
    x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn_like(x)
    step = timed(lambda: m(x).backward(w), 3)
    def infer():
        with torch.no_grad():
            m(x)
    fwd = timed(infer)
    path = ops.LAST_PATH["edgewise_fwd"]
    line = f"{gate_mode:8s} {'k3' if use_k3 else '  '}{' lens' if lens else ''} B={B}: train step {step:7.1f} ms | inference fwd {fwd:6.2f} ms (path {path})"
    if not use_k3:
        ops.set_path("generic")
        line += f" | generic path: train step {timed(lambda: m(x).backward(w), 3):6.1f} ms, inference fwd {timed(infer):6.2f} ms"
        ops.set_path("auto")
    print(line + f" | peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)


run("lowrank", False)
run("lowrank", False, lens=True)
run("dense", False)
run("dense", True)
