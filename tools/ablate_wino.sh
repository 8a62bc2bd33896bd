#!/bin/bash
# Build ablated variants of the Winograd kernel (timing only - results are wrong by construction) and time the layer shapes.
# usage (on the GPU box): bash tools/ablate_wino.sh
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
for abl in ${ABLS:-"" NOLOAD NOLDS NOMFMA}; do
  tag=$(echo "$abl" | tr -d ' -' ); tag=${tag:-full}
  out=/tmp/libssecg_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC ${abl:+-DSSECG_ABL_$abl} -shared \
      $SRC/conv.hip $SRC/conv_wino.hip $SRC/conv_wino4.hip $SRC/stem.hip $SRC/elementwise.hip $SRC/loss_optim.hip $SRC/augment.hip $SRC/amp.hip -o $out
  echo "== $tag"
  SSECG_LIB=$out SSECG_WINOGRAD=1 timeout -k 10 120 python tools/conv_bench.py fwd 1024 10 "${SHAPE:-k3   }" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_WINOGRAD
done
