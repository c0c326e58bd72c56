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
