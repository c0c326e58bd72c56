#!/bin/bash
# dev tool (GPU box): rebuild with s_memtime stamps for launch $1 (A|B|C) of the split backward and print the per-section cycles
export MOPK_STAMPS=${2:-1} MOPK_STAMP_PH=${1:-C}
python -c "from mop_amd import build; build.build_lib(force=True)" > /dev/null 2>&1 || exit 1
python tools/stamps.py 256 2>&1 | grep -v amdgpu.ids
