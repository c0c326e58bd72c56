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


def check_grads(grads, gref, gtol, scalar_tol=None, floor=1e-3, d=None, noise_factor=4.0):
    """per-parameter max-abs error / max|ref_k|.

    bf16 runs against a reference fixture `d` that carries the REFERENCE's own all-bfloat16 error per gradient tensor
    ('bf16err:<name>', tools/gen_golden.py): the limit for tensor k is max(gtol, noise_factor x that error) -- a gradient that is a
    small difference of large sums (gate-head weights at N = 197: the reference's bf16 run is 19-32 % off its own fp32 run) cannot be
    resolved by bf16 arithmetic, and where it can, gtol applies as it stands (a tensor also passes when its absolute error is below
    gtol x floor x the module's largest gradient: negligible at the module's scale).  (The recorded error is ONE sample of bf16 rounding noise and
    a run here is another: the factor covers the spread between two samples' maxima.)  No magnitude floor on this path.

    Without such a record (oracle-based sweeps, fp32 runs): the error is normalised by max(|ref_k|max, floor x the largest gradient
    magnitude of the module), which keeps analytically-zero / cancellation-dominated tensors from being judged against their own noise."""
    gscale = max(float(np.abs(v).max()) for v in gref.values())
    assert set(grads) == set(gref)
    for k in gref:
        g = np.asarray(grads[k]).reshape(gref[k].shape)
        noise = d.get("bf16err:" + k) if d is not None else None
        if noise is not None:
            lim = max(gtol, noise_factor * float(noise))
            err = max_abs(g, gref[k]) / max(float(np.abs(gref[k]).max()), 1e-30)
            # second clause: a tensor whose absolute error is below gtol x floor x the module's largest gradient is negligible at the
            # module's scale whatever its own size (analytic zeros, e.g. every gate gradient at N = 1; tensors 1e-4 of the module's scale)
            assert err <= lim or max_abs(g, gref[k]) <= gtol * floor * gscale, \
                f"{k} {err:.3e} (limit {lim:.1e}, bf16 noise sample {float(noise):.1e}; abs {max_abs(g, gref[k]):.2e} vs module scale {gscale:.2e})"
            continue
        den = max(float(np.abs(gref[k]).max()), floor * gscale, 1e-30)
        lim = scalar_tol if (scalar_tol is not None and gref[k].size == 1) else gtol
        assert max_abs(g, gref[k]) / den <= lim, f"{k} {max_abs(g, gref[k]) / den:.3e}"


def oracle_bf16_noise(module_fwd, module_bwd, x, w, params, *fwd_args, samples=4):
    """bf16 noise floor for oracle-based tests: the float64 oracle's own gradients move by this much (max-abs / max|exact|, per tensor;
    largest of `samples` draws) when its inputs -- x, the upstream gradient and every parameter -- are rounded to bfloat16 (draw 0:
    round-to-nearest-even; further draws: stochastic rounding, i.e. independent samples of the same rounding noise).  Returned in the
    'bf16err:<name>' form check_grads() reads, so a kernel that computes in bf16 is held to max(gtol, factor x this) instead of a
    magnitude floor.  Cancellation-dominated gradients (a scalar that is the sum of 10^4 signed terms 4000 x its size) get the wide
    limit they need, well-conditioned ones keep gtol."""
    rng = np.random.default_rng(1234)

    def bf16(a, stochastic):
        a = np.ascontiguousarray(a, dtype=np.float32)
        u = a.view(np.uint32).astype(np.uint64)
        add = rng.integers(0, 1 << 16, size=a.shape, dtype=np.uint64) if stochastic else (((u >> 16) & 1) + 0x7fff)
        return ((u + add) & 0xffff0000).astype(np.uint32).view(np.float32).astype(np.float64)
    out, cache = module_fwd(np.asarray(x, np.float64), params, *fwd_args)
    dx0, g0 = module_bwd(np.asarray(w, np.float64), cache)
    noise = {}
    for i in range(samples):
        pb = {k: bf16(v, i > 0) for k, v in params.items()}
        outb, cacheb = module_fwd(bf16(x, i > 0), pb, *fwd_args)
        dxb, gb = module_bwd(bf16(w, i > 0), cacheb)
        cur = {"bf16err:dx": max_abs(dxb, dx0) / max(float(np.abs(dx0).max()), 1e-30)}
        for k in g0:
            cur["bf16err:" + k] = max_abs(gb[k], g0[k]) / max(float(np.abs(g0[k]).max()), 1e-30)
        noise = {k: max(v, noise.get(k, 0.0)) for k, v in cur.items()}
    return noise
