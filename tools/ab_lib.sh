#!/bin/bash
# A/B of two library builds in ONE gpurun call (boxes differ by several % in sustained clock): the committed HEAD version
# of csrc/ vs the working tree.  usage: bash tools/ab_lib.sh <conv_bench args...>
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head/csrc /tmp/ab_head/include
for f in conv.hip conv_wino.hip elementwise.hip loss_optim.hip augment.hip; do cp .ab_head/$f /tmp/ab_head/csrc/; done
cp .ab_head/ssecg.h /tmp/ab_head/include/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/tmp/ab_head/include -shared /tmp/ab_head/csrc/*.hip -o /tmp/libssecg_head.so
for rep in 1 2; do
  echo "== HEAD (A), pass $rep"; SSECG_LIB=/tmp/libssecg_head.so python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_
  echo "== working tree (B), pass $rep"; python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_
done
