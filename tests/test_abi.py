"""The C-ABI library builds for gfx950, loads, and exports every symbol include/mopk.h declares."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from mop_amd import build
    build.build_lib()
    from mop_amd import _lib
    return _lib.lib()


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mopk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mopk_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from mop_amd import _lib
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mopk.h but not exported by libmopk.so"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in mop_amd/_lib.py"


def test_version_and_strerror(lib):
    assert lib.mopk_version() == 118
    assert lib.mopk_strerror(0) == b"ok"
    assert b"shape" in lib.mopk_strerror(-1)


def test_struct_sizes_match_header(lib):
    # sizes computed by the C compiler from the same header
    import subprocess, tempfile
    prog = r'''
#include <stdio.h>
#include "mopk.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(MopkView4), sizeof(MopkView5),
 sizeof(MopkEdgewiseArgs), sizeof(MopkDualPathArgs), sizeof(MopkQuartetArgs), sizeof(MopkSdpaArgs), sizeof(MopkEdgewiseExt),
 sizeof(MopkCrossViewArgs), sizeof(MopkLayerNormArgs), sizeof(MopkLensMeansArgs));return 0;}
'''
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "s.c")
        open(cpath, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cpath, "-o", exe])
        sizes = list(map(int, subprocess.check_output([exe]).split()))
    from mop_amd import _lib
    mine = [C.sizeof(t) for t in (_lib.View4, _lib.View5, _lib.EdgewiseArgs, _lib.DualPathArgs,
                                  _lib.QuartetArgs, _lib.SdpaArgs, _lib.EdgewiseExt, _lib.CrossViewArgs, _lib.LayerNormArgs,
                                  _lib.LensMeansArgs)]
    assert mine == sizes


def test_size_queries_need_no_gpu(lib):
    from mop_amd import _lib
    a = _lib.EdgewiseArgs()
    a.B, a.H, a.N, a.dk, a.V, a.r = 2, 6, 197, 64, 5, 4
    s = lib.mopk_edgewise_saved_bytes(C.byref(a))
    w = lib.mopk_edgewise_workspace_bytes(C.byref(a))
    assert s > 0 and w > 0
    a.N = 0
    assert lib.mopk_edgewise_saved_bytes(C.byref(a)) == 0


def test_lens_means_support_query_needs_no_gpu(lib):
    """mopk_lens_means_supported: shape gating of the closed-form lens kernels (N <= 224, dk 16 / 32 / 64, L <= 4, L V <= 16, LDS budget)"""
    from mop_amd import _lib
    a = _lib.LensMeansArgs()
    a.B, a.H, a.N, a.dk, a.V, a.L = 2, 6, 197, 64, 5, 2
    a.dil[0], a.dil[1] = 1, 2
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 1 and lib.mopk_lens_means_supported(C.byref(a), 1) == 1
    a.N = 225
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 0
    a.N, a.dk = 197, 48
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 0
    a.dk, a.V, a.L = 64, 5, 4                      # L V = 20 channels
    for i in range(4):
        a.dil[i] = 1
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 0
    a.V, a.L, a.N = 4, 4, 224                      # 16 channels: the backward's working set grows with the largest dilation
    assert lib.mopk_lens_means_supported(C.byref(a), 1) == 1
    a.dil[3] = 200
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 1 and lib.mopk_lens_means_supported(C.byref(a), 1) == 0
    a.dil[3] = 0
    assert lib.mopk_lens_means_supported(C.byref(a), 0) == 0
    assert lib.mopk_lens_means_fwd(None, None) < 0 and lib.mopk_lens_means_bwd(None, None) < 0


def test_bad_args_are_rejected_before_any_launch(lib):
    from mop_amd import _lib
    a = _lib.EdgewiseArgs()
    assert lib.mopk_edgewise_lowrank_fwd(C.byref(a), None) == -1  # bad shape
    a.B, a.H, a.N, a.dk, a.V, a.r = 1, 1, 8, 16, 2, 2
    assert lib.mopk_edgewise_lowrank_fwd(C.byref(a), None) == -2  # null pointers
