import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def load_golden(name):
    import numpy as np

    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    params = {k[6:]: v for k, v in d.items() if k.startswith("param:")}
    grads = {k[5:]: v for k, v in d.items() if k.startswith("grad:")}
    meta = {k[5:]: (v.item() if v.ndim == 0 else v) for k, v in d.items() if k.startswith("meta:")}
    return d, params, grads, meta


def ref_bf16_error(d, key):
    """max-abs error / max|fp32| of the REFERENCE's own all-bfloat16 run against its float32 run for gradient `key` ('dx' or a
    parameter name), recorded by tools/gen_golden.py; 0.0 when the fixture does not carry it."""
    v = d.get("bf16err:" + key)
    return float(v) if v is not None else 0.0
