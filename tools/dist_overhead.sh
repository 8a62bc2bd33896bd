#!/bin/bash
# What does the distributed machinery cost on ONE rank?  bench.py with a world-size-1 RCCL group, the data-parallel wrapper and forced
# SyncBN all-reduces (SSECG_BENCH_FORCE_DIST=1) under a rocprofv3 kernel trace, next to the plain run.  Reported per STEADY-STATE step
# (the launches between the last optimiser launches of the trace; construction-time broadcasts and first-step allocations excluded):
# launches, device-busy time, idle time (where the stream hand-offs to ProcessGroupNCCL's stream sit), and the per-kernel difference.
# usage (GPU box): bash tools/dist_overhead.sh <outdir>
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/dist_overhead}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in plain forced; do
  [ $mode = forced ] && export SSECG_BENCH_FORCE_DIST=1 || unset SSECG_BENCH_FORCE_DIST
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-amp-record > $OUT/$mode.log 2>&1
  cp $(find $OUT/$mode -name "*kernel_trace.csv" | head -1) $OUT/${mode}_kernel_trace.csv
  rm -rf $OUT/$mode
done
python3 - $OUT <<'PY'
import collections, csv, json, re, sys
out = sys.argv[1]
res = {}
for mode in ("plain", "forced"):
    rows = sorted(csv.DictReader(open(f"{out}/{mode}_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
    names = [re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:80] for r in rows]
    idx = [i for i, n in enumerate(names) if n.startswith("adamw_multi")]
    nsteps = 4                                     # four timed steps (the trace's last step is bench.py's instrumented one: skipped)
    lo, hi = idx[-2 - nsteps] + 1, idx[-2] + 1
    end, busy, idle, gaps = int(rows[lo - 1]["End_Timestamp"]), 0, 0, 0
    t0 = end
    per = collections.defaultdict(lambda: [0, 0.0])
    for i in range(lo, hi):
        st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        g = max(st - end, 0)
        idle += g; gaps += g > 5000
        busy += max(en - max(st, end), 0)
        end = max(end, en)
        per[names[i]][0] += 1; per[names[i]][1] += (en - st) / 1e3
    line = [l for l in open(f"{out}/{mode}.log") if l.startswith("{")][-1]
    res[mode] = dict(launches=(hi - lo) / nsteps, span=(end - t0) / nsteps / 1e6, busy=busy / nsteps / 1e6, idle=idle / nsteps / 1e6, gaps=gaps / nsteps,
                     per={k: (v[0] / nsteps, v[1] / nsteps) for k, v in per.items()}, ms=json.loads(line)["ms_per_step"])
p, f = res["plain"], res["forced"]
print(f"bench ms/step (untraced timing region of the same runs): plain {p['ms']:.3f}, forced {f['ms']:.3f} (+{f['ms'] - p['ms']:.3f})")
for k, r in res.items():
    print(f"{k:6s}: {r['launches']:.1f} launches/step, step span {r['span']:.3f} ms = device busy {r['busy']:.3f} + idle {r['idle']:.3f} ms ({r['gaps']:.0f} gaps > 5 us)")
for k in sorted(set(p["per"]) | set(f["per"]), key=lambda k: -(f["per"].get(k, (0, 0))[1] - p["per"].get(k, (0, 0))[1])):
    dn, dt = f["per"].get(k, (0, 0))[0] - p["per"].get(k, (0, 0))[0], f["per"].get(k, (0, 0))[1] - p["per"].get(k, (0, 0))[1]
    if abs(dn) >= 0.5 or abs(dt) > 20:
        print(f"  {dn:+7.1f} launches  {dt:+9.1f} us/step  {k}")
PY
rm -f $OUT/*_kernel_trace.csv
