#!/bin/bash
# Ablated builds of the stem forward kernel (timing only - results are wrong by construction).
# usage (on the GPU box): ABLS="full;NOSTORE;NOLOAD;NOMFMA" bash tools/ablate_stem.sh
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${ABLS:-full;NOSTORE;NOLOAD;NOMFMA}"
for abl in "${LIST[@]}"; do
  tag=$(echo "$abl" | tr " " "+")
  defs=""; for a in $abl; do [ "$a" = full ] || defs="$defs -DSSECG_ABLS_$a"; done
  out=/tmp/libssecg_stem_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $defs -shared \
      $SRC/conv.hip $SRC/conv_wino.hip $SRC/conv_wino4.hip $SRC/stem.hip $SRC/elementwise.hip $SRC/loss_optim.hip $SRC/augment.hip $SRC/amp.hip -o $out
  echo "== $tag"
  SSECG_LIB=$out timeout -k 10 120 python tools/stem_bench.py 1024 2>&1 | grep "stem.hip"
done
