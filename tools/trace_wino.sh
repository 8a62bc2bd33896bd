#!/bin/bash
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC -DSSECG_WINO_TRACE -shared \
    $SRC/conv.hip $SRC/conv_wino.hip $SRC/elementwise.hip $SRC/loss_optim.hip $SRC/augment.hip -o /tmp/libssecg_trace.so
SSECG_LIB=/tmp/libssecg_trace.so python tools/trace_wino.py "$@"
