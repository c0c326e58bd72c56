"""-m gpu: HIP-graph capture of a model built on the fused kernels (torch.cuda.make_graphed_callables), in its own process: the kernels
are stream-ordered, allocate nothing and set their one piece of per-kernel state (the dynamic-LDS attribute) before capture, so a
replay must reproduce the eager gradients bit for bit.  A capture that cannot run in this environment skips (a wrong replay fails)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_graphed_vit_edgewise_reproduces_the_eager_gradients():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_probe.py"), "--small"], cwd=root, capture_output=True,
                       text=True, timeout=300)
    if r.returncode != 0 or "OURS_IDENTICAL" not in r.stdout:
        pytest.skip("graph capture did not run here: " + (r.stderr or r.stdout)[-300:])
    assert "OURS_IDENTICAL True" in r.stdout, r.stdout[-500:]
