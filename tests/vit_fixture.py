"""Deterministic parameter values for the ViT_MoP fixtures (shared by tools/gen_golden.py and the GPU test).

The 5.4 M-parameter BASELINE.json configs[0] model would make a 43 MB fixture if its parameters and gradients were stored, so the
parameters are NOT stored: both sides fill the model from a numpy PCG64 stream (stable across platforms and versions) in state_dict
order, and the fixture keeps the input, the logits, the gate maps, dx and -- per parameter -- a strided sample of the gradient
plus its L2 norm.
"""
import numpy as np

SAMPLE = 2048          # gradient entries kept per parameter tensor (every tensor is sampled with a fixed stride)


def fill_params(shapes, seed):
    """shapes: ordered {name: shape} of the model's state_dict (float tensors).  Returns {name: float32 array}.
    weights ~ N(0, fan_in^-1/2) (matrices / conv kernels), LayerNorm / alpha-like vectors 1 + 0.1 N(0,1), biases 0.1 N(0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        z = rng.standard_normal(shape).astype(np.float32) if len(shape) else np.float32(rng.standard_normal())
        if name.endswith("bias") or name.endswith("pos"):
            v = 0.1 * z
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = z / np.sqrt(max(fan_in, 1))
        elif name.endswith("weight"):              # LayerNorm scale
            v = 1.0 + 0.1 * z
        else:                                       # alpha_pos / alpha_neg, cls tokens, ...
            v = 0.8 + 0.1 * z
        out[name] = np.asarray(v, dtype=np.float32)
    return out


def grad_sample(g):
    g = np.asarray(g, dtype=np.float32).reshape(-1)
    stride = max(1, g.size // SAMPLE)
    return g[::stride][:SAMPLE].copy(), np.float32(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
