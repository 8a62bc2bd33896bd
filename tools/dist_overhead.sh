#!/bin/bash
# What does the distributed machinery cost on ONE rank?  bench.py with a world-size-1 RCCL group, DDP and forced SyncBN all-reduces
# (SSECG_BENCH_FORCE_DIST=1) under rocprofv3 kernel stats, next to the plain run: extra kernels, their time, and the idle time
# (wall - sum of kernel time) the stream hand-offs between this library's launches and ProcessGroupNCCL's stream add.
# usage (GPU box): bash tools/dist_overhead.sh <outdir>
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/dist_overhead}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in plain forced; do
  [ $mode = forced ] && export SSECG_BENCH_FORCE_DIST=1 || unset SSECG_BENCH_FORCE_DIST
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-amp-record > $OUT/$mode.log 2>&1
  cp $(find $OUT/$mode -name "*kernel_stats.csv" | head -1) $OUT/${mode}_kernel_stats.csv
  rm -rf $OUT/$mode
done
python3 - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
res = {}
for mode in ("plain", "forced"):
    rows = list(csv.DictReader(open(f"{out}/{mode}_kernel_stats.csv")))
    calls = {r["Name"]: (int(r["Calls"]) / 8, float(r["TotalDurationNs"]) / 8 / 1e3) for r in rows}
    line = [l for l in open(f"{out}/{mode}.log") if l.startswith("{")][-1]
    res[mode] = (calls, json.loads(line)["ms_per_step"])
pc, pm = res["plain"]; fc, fm = res["forced"]
print(f"ms/step: plain {pm:.3f}, forced {fm:.3f} (+{fm - pm:.3f}); kernel time per step: plain {sum(v[1] for v in pc.values()) / 1e3:.3f} ms, forced {sum(v[1] for v in fc.values()) / 1e3:.3f} ms; "
      f"launches per step: plain {sum(v[0] for v in pc.values()):.0f}, forced {sum(v[0] for v in fc.values()):.0f}")
for k in sorted(set(pc) | set(fc), key=lambda k: -(fc.get(k, (0, 0))[1] - pc.get(k, (0, 0))[1])):
    dn, dt = fc.get(k, (0, 0))[0] - pc.get(k, (0, 0))[0], fc.get(k, (0, 0))[1] - pc.get(k, (0, 0))[1]
    if abs(dn) >= 0.5 or abs(dt) > 20:
        print(f"  {dn:+7.1f} launches  {dt:+9.1f} us/step  {k[:100]}")
PY
