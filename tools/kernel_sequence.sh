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
lo, hi = (idx[-3] + 1, idx[-2] + 1) if len(idx) >= 3 else (0, len(names))
# gap = idle time of the device before the launch (start - latest end so far): where stream hand-offs / host stalls sit
end, busy, gaps = int(rows[lo - 1]["End_Timestamp"]) if lo else int(rows[0]["Start_Timestamp"]), 0, []
for i in range(lo, hi):
    st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = max(st - end, 0)
    busy += max(en - max(st, end), 0)
    gaps.append(gap)
    end = max(end, en)
span = end - (int(rows[lo - 1]["End_Timestamp"]) if lo else int(rows[0]["Start_Timestamp"]))
print(f"# step span {span / 1e6:.3f} ms, device busy {busy / 1e6:.3f} ms, idle {sum(gaps) / 1e6:.3f} ms in {hi - lo} launches; "
      f"gaps > 5 us: {sum(1 for g in gaps if g > 5000)} totalling {sum(g for g in gaps if g > 5000) / 1e6:.3f} ms")
for i in range(lo, hi):
    g = gaps[i - lo]
    dur = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    print(i - lo, f"{dur:8.1f} us", names[i], f"  <-- {g / 1e3:.1f} us idle before" if g > 5000 else "")
PY
rm -rf $D $D.log
