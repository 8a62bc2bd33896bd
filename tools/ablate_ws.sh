#!/bin/bash
# Ablated builds of the weights-stationary bf16 conv kernel (csrc/amp_ws.hip): every variant issues the SAME instruction stream
# and the same number of vector-memory operations per stage (stores go out of range, DMA pieces load the zero constant), so the
# differences are the memory system's, not the schedule's.  Timing only - results are wrong by construction.
# usage (GPU box): bash tools/ablate_ws.sh   [ABLS="full;NOSTORE;..."] [EXTRA="-DWS_AHEAD_MAX=3"]
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${ABLS:-full;NOSTORE;NODMA;NOMFMA;NODMA NOSTORE;NODMA NOSTORE NOMFMA}"
for abl in "${LIST[@]}"; do
  tag=$(echo "$abl" | tr " " "+")
  defs=""; for a in $abl; do [ "$a" = full ] || defs="$defs -DSSECG_ABLW_$a"; done
  out=/tmp/libssecgw_$tag.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $defs $EXTRA -c $SRC/amp_ws.hip -o /tmp/amp_ws_$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $SRC/conv.o $SRC/conv_wino.o $SRC/conv_wino4.o $SRC/stem.o $SRC/elementwise.o $SRC/loss_optim.o $SRC/augment.o $SRC/amp.o /tmp/amp_ws_$tag.o -o $out
  echo "== $tag $EXTRA"
  SSECG_LIB=$out WS_LAYERS=${WS_LAYERS:-64x500x64,128x250x128,256x125x256} timeout -k 10 120 python tools/ws_bench.py 1024 2>&1 | grep -v amdgpu.ids
done
