#!/bin/bash
# In-kernel s_memtime shares of the weights-stationary conv kernel (diagnostic build; read the SHARES, not the length).
set -e
cd "$(dirname "$0")/.."
SRC=semi-seg-ecg_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC -DSSECG_WS_STAMP $EXTRA -c $SRC/amp_ws.hip -o /tmp/amp_ws_stamp.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $SRC/conv.o $SRC/conv_wino.o $SRC/conv_wino4.o $SRC/stem.o $SRC/elementwise.o $SRC/loss_optim.o $SRC/augment.o $SRC/amp.o /tmp/amp_ws_stamp.o -o /tmp/libssecgw_stamp.so
SSECG_LIB=/tmp/libssecgw_stamp.so timeout -k 10 120 python tools/stamp_ws.py 2>&1 | grep -v amdgpu.ids
