"""dev tool (GPU box): does one ViTEdgewise forward+backward capture into a HIP graph (torch.cuda.make_graphed_callables) and replay
with the same results?  Prints eager vs graphed step times."""
import sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from mop_amd.nn import ViTEdgewise
torch.manual_seed(0)
model = ViTEdgewise(dim=256, depth=8, heads=4, n_classes=100, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4,
                    gate_init="mix5", drop_path=0.0).cuda().to(torch.bfloat16)
x = torch.randn(256, 3, 32, 32, device="cuda", dtype=torch.bfloat16)
y = torch.randint(0, 100, (256,), device="cuda")
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

print("eager   %.2f ms/step" % timed(model), flush=True)
g = torch.cuda.make_graphed_callables(model, (x,))
print("captured", flush=True)
print("graphed %.2f ms/step" % timed(g), flush=True)
# same numbers?  one eager and one graphed step from identical parameters
import copy
sd = copy.deepcopy(model.state_dict())
def one(m):
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    loss = F.cross_entropy(m(x).float(), y)
    loss.backward()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
le, ge = one(model)
lg, gg = one(g)
print("loss eager %.6f graphed %.6f ; grads bit-identical: %s" % (float(le), float(lg), all(torch.equal(ge[k], gg[k]) for k in ge)), flush=True)
lg2, gg2 = one(g)
le2, ge2 = one(model)
print("graphed twice identical:", all(torch.equal(gg[k], gg2[k]) for k in gg), "| eager twice identical:", all(torch.equal(ge[k], ge2[k]) for k in ge))
worst = sorted(((float((ge[k].float() - gg[k].float()).abs().max() / ge[k].float().abs().max().clamp_min(1e-12)), k) for k in ge), reverse=True)[:6]
print("largest relative differences eager vs graphed:", [(f"{v:.1e}", k) for v, k in worst])
