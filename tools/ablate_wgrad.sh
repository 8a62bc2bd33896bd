#!/bin/bash
# Ablated builds of the bf16 weight-gradient ring kernel (csrc/amp.hip: conv_wgrad_b16s1_kernel): same instruction stream, parts of
# the work removed (DMA pieces read a cached constant / LDS fragment reads halved or dropped / MFMAs dropped).  Timing only - results
# are wrong by construction.   usage (GPU box): bash tools/ablate_wgrad.sh   [ABLS="full;NODMA;..."]
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${ABLS:-full;NODMA;HALFREAD;NOREAD;NOMFMA;NODMA NOREAD;NODMA NOMFMA;NODMA NOREAD NOMFMA}"
for abl in "${LIST[@]}"; do
  tag=$(echo "$abl" | tr " " "+")
  defs=""; for a in $abl; do [ "$a" = full ] || defs="$defs -DSSECG_ABLG_$a"; done
  out=/tmp/libssecgg_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $defs $EXTRA -c $SRC/amp.hip -o /tmp/amp_$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $SRC/conv.o $SRC/conv_wino.o $SRC/conv_wino4.o $SRC/stem.o $SRC/elementwise.o $SRC/loss_optim.o $SRC/augment.o /tmp/amp_$tag.o $SRC/amp_ws.o -o $out
  echo "== $tag $EXTRA"
  SSECG_LIB=$out WS_LAYERS=${WS_LAYERS:-128x250x128,256x125x256,512x63x512} timeout -k 10 120 python tools/wgrad_bench.py 1024 2>&1 | grep -v amdgpu.ids
done
