"""-m gpu: HIP-graph capture of a model built on the fused kernels (torch.cuda.make_graphed_callables), in its own process: the kernels
are stream-ordered, allocate nothing and set their one piece of per-kernel state (the dynamic-LDS attribute) before capture, so a
replay must reproduce the eager gradients bit for bit.  Only an explicit CAPTURE_UNSUPPORTED line from the probe skips; a crash fails."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_graphed_vit_edgewise_reproduces_the_eager_gradients():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_probe.py"), "--small"], cwd=root, capture_output=True,
                       text=True, timeout=300)
    # Only an explicit refusal to capture skips.  Any abnormal end of the probe -- a signal (negative return code: the segmentation fault in
    # capture_end this test once hid), a Python error, a missing verdict line -- fails.
    if r.returncode == 0 and "CAPTURE_UNSUPPORTED" in r.stdout:
        pytest.skip("graph capture refused here: " + r.stdout[-300:])
    assert r.returncode == 0, f"graph probe ended abnormally (rc {r.returncode}): " + (r.stderr or r.stdout)[-600:]
    assert "OURS_IDENTICAL True" in r.stdout, r.stdout[-500:]
