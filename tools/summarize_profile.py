#!/usr/bin/env python3
"""Turn rocprofv3 outputs copied back under gpurun_out/ into the markdown summaries kept in profiles/.
usage: summarize_profile.py <kernel_stats.csv> <fetch counter csv> <write counter csv> <steps in trace> <out.md> <title> [B C L dtype]

The ``*_traffic.json`` written beside the summary carries the sha256 of the kernel sources it was measured on
(``bench.kernel_source_hash``) and the workload (B, C, L, dtype): bench.py only quotes it for exactly that build + workload."""
import collections
import csv
import re
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k)
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    return agg


def main():
    stats, fetch, write, steps, out, title = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
    f, w = load(fetch, 'FETCH_SIZE'), load(write, 'WRITE_SIZE')
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    amp = len(sys.argv) > 10 and sys.argv[10] == "bf16"
    extra = " --amp" if amp else ""
    o = [f"# {title}\n",
         f"`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps - 3} --warmup 2 --no-cpu-baseline --no-amp-record{extra}` ({steps} FixMatch steps in the",
         f"trace: warm-up + timed + 1 instrumented; B=512/GPU, 12 leads, L=2000, {'bf16 student pass (use_amp), fp32 teacher' if amp else 'fp32'}, one MI355X) with",
         "`SSECG_OVERLAP_PASSES=0` exported (the step on ONE stream, as in bench.py's instrumented step that the `roofline` record is taken from: in the",
         "default step the pseudo-label pass runs on a side stream beside the student forward and launches of the two passes share the chip, so their",
         "durations are not per-kernel quantities; the trace of the default two-stream step is the `*_two_streams.csv` beside this file), plus separate",
         f"`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of `bench.py --steps 1 --warmup 1{extra}` (counter collection serialises the launches).",
         "HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF of a wide coalesced read",
         "stream (MI355X_MICROARCH.md, HBM section), so read MB = 2 * FETCH_SIZE * 1024 / 1e6; WRITE_SIZE * 1024 is exact for 16-B",
         "stores. Infinity-Cache hits are included in both (they count L2 fabric requests).\n",
         f"Total kernel time {tot / steps / 1e6:.2f} ms/step.\n",
         "| ms/step | % | launches/step | avg us | read MB/launch (corrected) | write MB/launch | kernel |", "|---|---|---|---|---|---|---|"]
    for r in rows[:36]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name']); key = re.sub(r'\(.*', '', n)
        fr = f[key][0] / max(f[key][1], 1) * 1024 * 2 / 1e6 if key in f else float('nan')
        wr = w[key][0] / max(w[key][1], 1) * 1024 / 1e6 if key in w else float('nan')
        o.append(f"| {float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['Percentage']):.1f} | {int(r['Calls']) / steps:.1f} | "
                 f"{float(r['AverageNs']) / 1e3:.1f} | {fr:.1f} | {wr:.1f} | `{n[:100]}` |")
    open(out, 'w').write("\n".join(o) + "\n")
    # machine-readable per-launch HBM traffic (bench.py fills roofline.traffic from it)
    import json
    traffic = {}
    for r in rows:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name']); key = re.sub(r'\(.*', '', n)
        if key in f and key in w:
            traffic[re.sub(r'^void ', '', key)] = {"read_bytes": f[key][0] / max(f[key][1], 1) * 1024 * 2,
                                                  "write_bytes": w[key][0] / max(w[key][1], 1) * 1024,
                                                  "avg_ns": float(r['AverageNs']), "launches_per_step": int(r['Calls']) / steps}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    wl = sys.argv[7:11]
    workload = {"B": int(wl[0]), "C": int(wl[1]), "L": int(wl[2]), "dtype": wl[3]} if len(wl) == 4 else {"B": 512, "C": 12, "L": 2000, "dtype": "f32"}
    json.dump({"source": out, "source_hash": kernel_source_hash(), "workload": workload, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE x2 on "
               "gfx950 (MI355X_MICROARCH.md); averages over the launches of each kernel in one bench step", "kernels": traffic},
              open(re.sub(r'\.md$', '', out) + "_traffic.json", 'w'), indent=1)
    print("\n".join(o[7:26]))


if __name__ == "__main__":
    main()
