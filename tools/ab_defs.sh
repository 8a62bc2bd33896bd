#!/bin/bash
# Same-box A/B of builds that differ by -D switches only.  usage (GPU box): DEFS="-DA;-DB -DC;" bash tools/ab_defs.sh <conv_bench args>
# (an empty entry = the default build); two passes each, interleaved.
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
IFS=";" read -ra LIST <<< "${DEFS:-;}"
i=0
for d in "${LIST[@]}"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $d -shared $SRC/*.hip -o /tmp/libssecg_v$i.so
done
for rep in 1 2; do
  i=0
  for d in "${LIST[@]}"; do
    i=$((i+1))
    echo "== [$d] pass $rep"
    SSECG_LIB=/tmp/libssecg_v$i.so python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_
  done
done
