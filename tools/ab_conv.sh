#!/bin/bash
# Same-box A/B of two builds of the library on the conv micro-benchmark (boxes differ by up to 15 % on the MFMA-bound kernels,
# so a candidate is only ever compared with a baseline timed in the same gpurun call).
# usage: bash tools/ab_conv.sh <baseline.so> [mode] [shape filter]
BASE=$1; MODE=${2:-fwd}; SHAPE=${3:-k3   }
cd "$(dirname "$0")/.."
for i in 1 2; do
  echo "-- baseline"; SSECG_LIB=$BASE python tools/conv_bench.py $MODE 1024 10 "$SHAPE" 2>&1 | grep -v amdgpu
  echo "-- candidate"; python tools/conv_bench.py $MODE 1024 10 "$SHAPE" 2>&1 | grep -v amdgpu
done
