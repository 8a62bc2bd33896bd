#!/bin/bash
# PMC passes over one Winograd conv launch shape (l4) - separate runs per counter group (no trace domains combined).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wino
mkdir -p $OUT
i=0
for grp in ${PMC_GROUPS:-"TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"}; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 150 rocprofv3 --pmc ${grp//,/ } --kernel-trace --output-format csv -d $OUT/g$i -o r -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py fwd 1024 3 "l4 " > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "  (no counter file; see $OUT/g$i.log)"; grep -m2 -i "error\|exceeds" $OUT/g$i.log; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_wino" in r["Kernel_Name"] and "weight" not in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k:45s} {v / n:16.0f} per launch ({n} launches)")
PY
done
