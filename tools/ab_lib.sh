#!/bin/bash
# A/B of two library builds in ONE gpurun call (boxes differ by several % in sustained clock): the committed HEAD version
# of csrc/ vs the working tree.  usage: bash tools/ab_lib.sh <conv_bench args...>
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--snapshot" ]; then   # run in the build container before gpurun
  rm -rf .ab_head && mkdir -p .ab_head
  for f in $(git ls-tree --name-only HEAD semi-seg-ecg_amd/csrc/ | grep -E "\.(hip|h)$"); do git show HEAD:$f > .ab_head/$(basename $f); done
  git show HEAD:include/ssecg.h > .ab_head/ssecg.h
  echo "snapshot of HEAD kernels in .ab_head/"; exit 0
fi
SRC=semi-seg-ecg_amd/csrc
# .ab_head/ = `tools/ab_lib.sh --snapshot` output: HEAD's csrc/*.hip, csrc/*.h and include/ssecg.h (git-ignored, travels)
rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head && cp .ab_head/* /tmp/ab_head/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/tmp/ab_head -shared /tmp/ab_head/*.hip -o /tmp/libssecg_head.so
for rep in 1 2; do
  echo "== HEAD (A), pass $rep"; SSECG_LIB=/tmp/libssecg_head.so python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_
  echo "== working tree (B), pass $rep"; python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v SSECG_
done
