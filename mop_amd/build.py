"""Build libmopk.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to an object in parallel; the two fused-kernel sources are compiled once per
(NT, DK) instantiation pair (-DMOPK_INST_NT/-DMOPK_INST_DK) plus once as the dispatcher (NT=0).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmopk.so")
PLAIN = ["api.hip", "edgewise_generic.hip", "attn_generic.hip", "sdpa_flash.hip", "quartet_flash.hip", "layernorm.hip", "lens_means.hip"]
FUSED = ["edgewise_fused.hip", "edgewise_fused_bwd.hip"]
FUSED_INST_ONLY = []      # files compiled per (NT, DK) only (entry points called from the files above)
NTS, DKS = (1, 2, 3, 4, 5, 6, 7), (16, 32, 64)      # NT = 3: N <= 96 (the reference's CIFAR sequence length, N = 65)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("MOPK_EXTRA_FLAGS"):  # compiler experiments, e.g. MOPK_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp"
    FLAGS += os.environ["MOPK_EXTRA_FLAGS"].split()
if os.environ.get("MOPK_WHATIF"):  # timing experiments (wrong results by construction), e.g. MOPK_WHATIF=NOBAR,NOSLOT
    FLAGS += ["-DMOPK_WHATIF_" + w for w in os.environ["MOPK_WHATIF"].split(",")]
if os.environ.get("MOPK_STAMPS"):  # diagnostic build: s_memtime stamps per phase (never benchmark this build)
    FLAGS.append("-DMOPK_STAMPS")
    if os.environ["MOPK_STAMPS"] == "2":  # plus sub-phase stamps (shifts the phase names of tools/stamps.py)
        FLAGS.append("-DMOPK_STAMPS2")
    # which launch of the split backward writes the stamps: A / B / C (default C)
    FLAGS.append("-DMOPK_STAMP_PH=" + {"A": "0", "B": "1", "C": "2"}[os.environ.get("MOPK_STAMP_PH", "C")])


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libmopk.so cannot be built")


def _jobs():
    jobs = [(s, [], s.replace(".hip", ".o")) for s in PLAIN]
    for s in FUSED + FUSED_INST_ONLY:
        stem = s.replace(".hip", "")
        # fused bf16-MFMA kernels: relaxed fp (reassociation, contraction, approximate reciprocals) but inf/nan kept;
        # the exact-fp32 generic path is built without it
        fm = ["-ffast-math", "-fno-finite-math-only"]
        if s not in FUSED_INST_ONLY:
            jobs.append((s, ["-DMOPK_INST_NT=0", "-DMOPK_INST_DK=0"], f"{stem}_disp.o"))
        for nt in NTS:
            for dk in DKS:
                jobs.append((s, fm + [f"-DMOPK_INST_NT={nt}", f"-DMOPK_INST_DK={dk}"], f"{stem}_nt{nt}_dk{dk}.o"))
    return jobs


def _deps_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mopk.h"),
                                                               os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def is_stale() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < _deps_mtime()


def build_lib(force: bool = False, verbose: bool = False, workers: int = 8) -> str:
    if not force and not is_stale():
        return LIB
    cc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    newest = _deps_mtime()

    def compile_one(job):
        src, defs, obj = job
        out = os.path.join(OBJ, obj)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            return out
        cmd = [cc] + FLAGS + defs + ["-c", "-o", out, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src} {defs}:\n" + r.stdout + r.stderr)
        return out

    # biggest kernels first so the pool stays busy
    jobs = sorted(_jobs(), key=lambda j: (-("nt7" in j[2]) * 2 - ("nt4" in j[2]), j[2]))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=False))
