"""The product package never imports, links or executes the oracle (or the reference), and has no CPU fallback."""
import os
import re

from conftest import ROOT


def _py_files(d):
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(r, f)


def test_product_never_touches_oracle_or_reference():
    bad = []
    for f in _py_files(os.path.join(ROOT, "mop_amd")):
        src = open(f).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "/root/reference" in src or "MOP_REFERENCE" in src:
            bad.append(f)
    assert not bad, bad
    for f in os.listdir(os.path.join(ROOT, "mop_amd", "csrc")):
        src = open(os.path.join(ROOT, "mop_amd", "csrc", f)).read()
        assert "oracle/" not in src.replace("oracle/edgewise.py::core_bwd", "").replace("oracle/{sdpa,multihop,quartet}.py", "").replace("oracle/edgewise.py", "")


def test_only_allowed_files_import_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, f)).read(), re.M), f
    for f in _py_files(os.path.join(ROOT, "tools")):
        if os.path.basename(f) in ("check_shape.py",):   # dev-only checker tool, not product
            continue
        assert not re.search(r"^\s*(from|import)\s+oracle\b", open(f).read(), re.M), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mop_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.lib()
