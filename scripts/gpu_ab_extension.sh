#!/bin/bash
# Same-box A/B of two builds of the HIP extension (boxes differ by several per cent on MFMA-dense
# kernels, so two builds must alternate on ONE box within one call to be compared).
#
#   on the CPU box:  build variant A, cp kvmini/_C.*.so ab/_C_A.so; build variant B, cp ... ab/_C_B.so
#   on the GPU box:  scripts/gpu_ab_extension.sh ab "python scripts/gpu_microbench.py --what attn"
#
# Runs the command A B A B with the respective shared object in place and restores the original.
set -e
dir=${1:?directory holding _C_A.so and _C_B.so}
shift
cmd=${1:?command to run under each build}
so=$(ls kvmini/_C.*.so | head -1)
cp "$so" "$dir/_C_keep.so"
trap 'cp "$dir/_C_keep.so" "$so"' EXIT
for v in A B A B; do
  cp "$dir/_C_$v.so" "$so"
  echo "== build $v"
  timeout -k 10 300 bash -c "$cmd"
done
