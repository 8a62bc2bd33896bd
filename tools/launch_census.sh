#!/bin/bash
# Launch census of one FixMatch step at a small-batch configuration (BASELINE config #2 shape: B = 256, 1 lead): kernel stats
# of bench.py under rocprofv3, launches per step and time per kernel.  usage (GPU box): bash tools/launch_census.sh <tag> [bench args]
set -u
TAG=${1:-census}
[ $# -gt 0 ] && shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-amp-record "$@" > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - "$OUT/kernel_stats.csv" 8 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{calls / steps:.1f} launches/step, {tot / steps / 1e6:.3f} ms of kernel time per step ({steps} steps in the trace)")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:25]:
    print(f"  {int(r['Calls']) / steps:6.1f} /step  {float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  {r['Name'][:90]}")
PY
tail -1 $OUT/stats.log | cut -c1-400
