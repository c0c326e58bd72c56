"""helpers shared by the -m gpu parity tests"""
import numpy as np
import torch


def module_from_golden(cls, params, **ctor):
    m = cls(**ctor)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def run_fwd_bwd(m, x, w, dtype=torch.float32, **fk):
    xt = torch.from_numpy(x).cuda().to(dtype).requires_grad_(True)
    if dtype != torch.float32:
        m = m.to(dtype)
    y = m(xt, **fk)
    y.backward(torch.from_numpy(w).cuda().to(dtype))
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    return y.detach().float().cpu().numpy(), xt.grad.detach().float().cpu().numpy(), grads


def max_abs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def rel_err(a, b):
    """max-abs error normalised by the largest reference magnitude (for gradients)."""
    return max_abs(a, b) / max(1e-12, float(np.abs(b).max()))


def check_grads(grads, gref, gtol, scalar_tol=None, floor=1e-3):
    """per-parameter max-abs error / max(|ref_k|max, floor * largest gradient magnitude of the module).
    The floor keeps cancellation-dominated tensors (e.g. a gate-head weight whose true gradient is 1e-4 of
    the others, or analytically-zero ones that hold fp32 noise in the reference) from being judged
    relative to their own noise; bf16 arithmetic cannot resolve them and neither can the reference's own bf16 run."""
    gscale = max(float(np.abs(v).max()) for v in gref.values())
    assert set(grads) == set(gref)
    for k in gref:
        g = np.asarray(grads[k]).reshape(gref[k].shape)
        den = max(float(np.abs(gref[k]).max()), floor * gscale, 1e-30)
        lim = scalar_tol if (scalar_tol is not None and gref[k].size == 1) else gtol
        assert max_abs(g, gref[k]) / den <= lim, f"{k} {max_abs(g, gref[k]) / den:.3e}"
