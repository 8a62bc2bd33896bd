#!/bin/bash
# Ablated builds of the F(4,3) Winograd kernel (timing only - results are wrong by construction) on the layer shapes.
# usage (on the GPU box): bash tools/ablate_wino4.sh
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${ABLS:-full;NOLOAD;NOSTORE;NOMFMA}"
for abl in "${LIST[@]}"; do   # an entry may name several ablations: "NOLOAD NOSTORE"
  tag=$(echo "$abl" | tr " " "+")
  defs=""; for a in $abl; do [ "$a" = full ] || defs="$defs -DSSECG_ABL4_$a"; done
  out=/tmp/libssecg4_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $defs -shared \
      $SRC/*.hip -o $out
  echo "== $tag"
  SSECG_LIB=$out SSECG_WINO_F=4 timeout -k 10 120 python tools/conv_bench.py fwd 1024 10 "${SHAPE:-k3   }" 2>&1 | grep -v amdgpu.ids
done
