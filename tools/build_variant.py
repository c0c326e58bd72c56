#!/usr/bin/env python3
"""dev tool: link mop_amd/libmopk_<name>.so from the objects of the regular build, with some fused-kernel instantiations recompiled
under extra flags (diagnostic / what-if builds of one kernel without a full rebuild).  Use with MOPK_LIB=<that file>.

    python tools/build_variant.py stamps "-DMOPK_STAMPS" edgewise_fused:7:64 [edgewise_fused_bwd:7:64 ...]
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mop_amd import build as b

name, extra = sys.argv[1], sys.argv[2].split()
swaps = {}
cc = b._hipcc()
b.build_lib()                                   # regular objects must exist
os.makedirs(os.path.join(b.OBJ, name), exist_ok=True)
for spec in sys.argv[3:]:
    stem, nt, dk = spec.split(":")
    obj = f"{stem}_nt{nt}_dk{dk}.o"
    out = os.path.join(b.OBJ, name, obj)
    base = [f for f in b.FLAGS if not (f.startswith("--offload-arch=") and any(e.startswith("--offload-arch=") for e in extra))]   # an arch in `extra` replaces the default one
    cmd = [cc] + base + ["-ffast-math", "-fno-finite-math-only", f"-DMOPK_INST_NT={nt}", f"-DMOPK_INST_DK={dk}"] + extra + \
          ["-c", "-o", out, os.path.join(b.CSRC, stem + ".hip")]
    subprocess.run(cmd, check=True)
    swaps[obj] = out
objs = []
for _, _, obj in b._jobs():
    objs.append(swaps.get(obj, os.path.join(b.OBJ, obj)))
lib = os.path.join(b.HERE, f"libmopk_{name}.so")
subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
