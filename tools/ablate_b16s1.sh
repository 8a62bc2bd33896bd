#!/bin/bash
# Ablated builds of the bf16 LDS-DMA ring kernel (timing only - results are wrong by construction): which of its three
# streams (DMA requests, MFMAs, output stores) sets the pace.  usage (GPU box): bash tools/ablate_b16s1.sh
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${ABLS:-full;NOSTORE;NOMFMA;NODMA;NODMAW;NODMA NOSTORE;NOMFMA NOSTORE}"
for abl in "${LIST[@]}"; do
  tag=$(echo "$abl" | tr " " "+")
  defs=""; for a in $abl; do [ "$a" = full ] || defs="$defs -DSSECG_ABLB_$a"; done
  out=/tmp/libssecgb_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $defs -shared $SRC/*.hip -o $out
  echo "== $tag"
  SSECG_LIB=$out timeout -k 10 120 python tools/amp_bench.py 1024 2>&1 | grep -v amdgpu.ids | grep -E "^ 128  250  128|^ 256  125  256|^ 512   63  512" | cut -c1-75
done
