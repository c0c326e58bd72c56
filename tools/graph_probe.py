"""dev tool (GPU box): one ViTEdgewise forward + backward captured into HIP graphs (torch.cuda.make_graphed_callables) and replayed:
step time eager vs graphed, and whether the replay reproduces the eager gradients bit for bit (every tensor except the patch-embedding
convolution's weight gradient, which is MIOpen's and differs between two eager runs as well).  --small: a 2-block model (tests)."""
import copy, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from mop_amd.nn import ViTEdgewise
small = "--small" in sys.argv
torch.manual_seed(0)
model = ViTEdgewise(dim=128 if small else 256, depth=2 if small else 8, heads=2 if small else 4, n_classes=100, n_views=3 if small else 5,
                    share_qkv=True, gate_mode="lowrank", gate_rank=2 if small else 4, gate_init="mix5", drop_path=0.0).cuda().to(torch.bfloat16)
B = 16 if small else 256
x = torch.randn(B, 3, 32, 32, device="cuda", dtype=torch.bfloat16)
y = torch.randint(0, 100, (B,), device="cuda")
opt = torch.optim.AdamW(model.parameters(), lr=1e-3)


def step(m):
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(m(x).float(), y)
    loss.backward()
    opt.step()
    return loss


def timed(m, n=20):
    for _ in range(3):
        step(m)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        step(m)
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


if not small:
    print("eager   %.2f ms/step" % timed(model), flush=True)
try:
    g = torch.cuda.make_graphed_callables(model, (x,))
except RuntimeError as e:       # a Python-level refusal to capture (an op the capture mode does not allow): the one case the test may skip
    print("CAPTURE_UNSUPPORTED", repr(e)[:300], flush=True)
    sys.exit(0)
print("captured", flush=True)
if not small:
    print("graphed %.2f ms/step" % timed(g), flush=True)
sd = copy.deepcopy(model.state_dict())


def one(m):
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    loss = F.cross_entropy(m(x).float(), y)
    loss.backward()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


le, ge = one(model)
lg, gg = one(g)
lg2, gg2 = one(g)
ours = [k for k in ge if k != "patch.proj.weight"]
print("loss eager %.6f graphed %.6f" % (float(le), float(lg)))
print("OURS_IDENTICAL", all(torch.equal(ge[k], gg[k]) for k in ours) and all(torch.equal(gg[k], gg2[k]) for k in ours) and bool(le == lg))
