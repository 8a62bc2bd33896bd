#!/bin/bash
# rocprofv3 kernel-trace durations of the bf16 conv kernels under tools/ws_bench.py (GPU-side start-to-end, per kernel name + grid)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ws_prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python3 $GRAFT_REPO_ROOT/tools/ws_bench.py 1024 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_b16" in k:
        k = re.sub(r"\(.*", "", k).replace("ssecg_amp::", "").replace("(anonymous namespace)::", "")
        acc[(k, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items()):
    v.sort()
    print(f"{k:50s} grid {g:>8s}  n {len(v):4d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
PY
