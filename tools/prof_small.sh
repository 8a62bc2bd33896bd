#!/bin/bash
# rocprofv3 kernel statistics of bench.py at a small batch (GPU box).  usage: bash tools/prof_small.sh <tag> <batch> [extra bench.py args]
TAG=$1; B=$2; shift; shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $REPO/bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-amp-record "$@" > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -1 $OUT/stats.log | cut -c1-200
python3 - $OUT/kernel_stats.csv 26 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.3f} ms/step over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
PY
