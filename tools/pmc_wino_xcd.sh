#!/bin/bash
# FETCH_SIZE / WRITE_SIZE and time of the Winograd forward kernel on the layer3 / layer4 shapes for both work mappings
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_xcd
mkdir -p $OUT
for xcd in 1 0; do
  for shape in "l3 " "l4 "; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=$OUT/x${xcd}_${shape// /}_$ctr
      SSECG_WINO_XCD=$xcd timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py fwd 1024 3 "$shape" > $d.log 2>&1
      python3 - "$(find $d -name '*counter_collection.csv' | head -1)" "xcd_map=$xcd $shape $ctr" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "conv_wino_kernel" in r["Kernel_Name"]]
mult = 2 if "FETCH" in sys.argv[2] else 1
print(f"{sys.argv[2]:34s} {sum(v) / len(v) * 1024 * mult / 1e6:9.1f} MB per launch ({len(v)} launches)")
PY
    done
    SSECG_WINO_XCD=$xcd python3 $GRAFT_REPO_ROOT/tools/conv_bench.py fwd 1024 10 "$shape" 2>&1 | grep -v amdgpu | grep -v SSECG
  done
done
