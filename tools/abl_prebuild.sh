#!/bin/bash
# Pre-build timing-only (ablated / experimental) variants of ONE kernel source HERE (hipcc cross-compiles gfx950 without a GPU) so that
# a GPU call spends its minutes measuring, not compiling: the variant's object is linked with the other, unchanged objects of csrc/ into
# build/abl/libssecg_<tag>.so (git-ignored; travels with the gpurun snapshot).  Run `make -C semi-seg-ecg_amd/csrc` first.
# usage: bash tools/abl_prebuild.sh <source.hip> "<tag>:<-D flags>" ["<tag2>:<flags2>" ...]     e.g.
#        bash tools/abl_prebuild.sh conv_wino4.hip "full:" "NOGSTORE:-DSSECG_ABL4_NOGSTORE"
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
FILE=$1; shift
BASE=${FILE%.hip}
mkdir -p build/abl
OTHERS=$(ls $SRC/*.o | grep -v "/$BASE.o$")
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude -I$SRC $defs -c $SRC/$FILE -o build/abl/${BASE}_$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/abl/${BASE}_$tag.o $OTHERS -o build/abl/libssecg_$tag.so
  echo "built build/abl/libssecg_$tag.so ($defs)"
done
