#!/bin/bash
# dev tool (GPU box): same-box A/B of two prebuilt libraries mop_amd/libmopk_A.so / libmopk_B.so (ABAB order);
# box-to-box variance of the core kernels (2-3 %) is larger than most single optimisations
for v in A B A B; do
  cp mop_amd/libmopk_$v.so mop_amd/libmopk.so
  echo "== $v"; timeout -k 10 200 python tools/bench_modes.py 2>&1 | tail -2
done
