#!/bin/bash
# Ordered kernel list of the LAST traced step of bench.py (names shortened): shows which launches surround the small fill /
# copy kernels.  usage (GPU box): bash tools/kernel_sequence.sh <outfile> [bench args]
set -u
OUTF=$1; shift
REPO=$GRAFT_REPO_ROOT
D=/tmp/kseq_$$; mkdir -p $(dirname $REPO/$OUTF)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $D -o r -- python3 $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-amp-record "$@" > $D.log 2>&1
python3 - $(find $D -name "*kernel_trace.csv" | head -1) > $REPO/$OUTF <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:70] for r in rows]
# last step = after the last adamw but one
idx = [i for i, n in enumerate(names) if n.startswith("adamw_multi")]
seq = names[idx[-3] + 1: idx[-2] + 1] if len(idx) >= 3 else names
for i, n in enumerate(seq):
    print(i, n)
PY
rm -rf $D $D.log
