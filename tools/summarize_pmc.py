#!/usr/bin/env python3
"""PMC counter passes (tools/pmc_kernel.sh output) -> profiles/<name>.md + the ``pmc`` entries of a *_traffic.json.

usage: summarize_pmc.py <pmc_kernel.sh output> <out.md> <title> [<traffic.json to merge into>]

Derived figures (counters are summed over the chip by rocprofv3; 8 XCDs x 32 CUs x 4 SIMDs = 1024 SIMDs):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)   share of SIMD-cycles (GRBM reference clock) the matrix pipe works
  waiting    = SQ_WAIT_ANY / SQ_WAVE_CYCLES                              wave-cycles parked at s_waitcnt / barriers
  lds_active = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
The traffic json carries the kernel-source hash bench.py checks before quoting ``roofline.traffic`` / ``roofline.mfma_busy``."""
import collections
import json
import os
import re
import sys


def parse(path):
    rows = collections.defaultdict(dict)       # (kernel, grid) -> {counter: (value per launch, launches)}
    pat = re.compile(r"^(?:void )?(.*?)\s+grid\s+(\d+)\s+(\S+)\s+(\d+) per launch \((\d+)\)")
    for line in open(path):
        m = pat.match(line.rstrip())
        if m:
            k, grid, ctr, val, n = m.group(1).strip(), int(m.group(2)), m.group(3), float(m.group(4)), int(m.group(5))
            rows[(k, grid)][ctr] = (val, n)
    return rows


def derive(c):
    g = lambda k: c[k][0] if k in c else None
    out = {}
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
        out["grbm_cycles_per_xcd"] = g("GRBM_GUI_ACTIVE") / 8.0
        out["mfma_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0)
    if g("SQ_WAIT_ANY") is not None and g("SQ_WAVE_CYCLES"):
        out["waiting"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE") is not None and g("SQ_BUSY_CYCLES"):
        out["lds_active"] = g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        out["lds_conflict"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_INSTS_VALU") is not None and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["valu_per_mfma_kcycle"] = g("SQ_INSTS_VALU") / (g("SQ_VALU_MFMA_BUSY_CYCLES") / 1000.0)
    return out


def main():
    src, out_md, title = sys.argv[1], sys.argv[2], sys.argv[3]
    merge = sys.argv[4] if len(sys.argv) > 4 else None
    rows = parse(src)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    h = kernel_source_hash()
    table = []
    for (k, grid), c in rows.items():
        d = derive(c)
        if "mfma_busy" not in d:
            continue
        n = max(v[1] for v in c.values())
        table.append((d["grbm_cycles_per_xcd"] * n, k, grid, n, d))
    table.sort(key=lambda t: -t[0])
    o = [f"# {title}\n",
         "`bash tools/pmc_kernel.sh conv_ -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amp-record` (four `rocprofv3 --pmc ... --kernel-trace`",
         "passes, the program directly after `--`; B = 512, 12 leads, L = 2000, one MI355X; averages per launch over the launches of the pass -",
         f"warm-up + timed + instrumented step - counters summed over the chip).  Kernel-source hash `{h}` (bench.py quotes `roofline.mfma_busy`",
         "only for this build).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): share of SIMD-cycles at the GRBM",
         "reference clock in which the matrix pipe works; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; LDS = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES and bank",
         "conflicts / LDS-active cycles; VALU : MFMA = SQ_INSTS_VALU per MFMA-busy kilo-cycle.\n",
         "| kernel | grid (threads) | launches in pass | GRBM cycles / XCD | MFMA busy | waves waiting | LDS active / SQ busy | bank conflicts / LDS active | VALU per MFMA kcycle |",
         "|---|---|---|---|---|---|---|---|---|"]
    f = lambda v, p=2: "-" if v is None else f"{v:.{p}f}"
    for _, k, grid, n, d in table:
        o.append(f"| `{k}` | {grid} | {n} | {d['grbm_cycles_per_xcd']:,.0f} | {f(d.get('mfma_busy'))} | {f(d.get('waiting'))} | {f(d.get('lds_active'))} | "
                 f"{f(d.get('lds_conflict'), 3)} | {f(d.get('valu_per_mfma_kcycle'), 1)} |")
    open(out_md, "w").write("\n".join(o) + "\n")
    print("\n".join(o[8:22]))
    if merge:
        tj = json.load(open(merge))
        if tj.get("source_hash") != h:
            raise SystemExit(f"{merge} was measured on kernel sources {tj.get('source_hash')}, the tree is {h}: not merged")
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for _, k, grid, n, d in table:                       # launch-weighted over the grids of one kernel
            for key in ("mfma_busy", "waiting", "lds_active", "lds_conflict"):
                if key in d:
                    per[k][key] += d[key] * n
            per[k]["_n"] += n
        for k, acc in per.items():
            if k in tj["kernels"]:
                tj["kernels"][k]["pmc"] = {key: v / acc["_n"] for key, v in acc.items() if key != "_n"}
        tj["pmc_source"] = out_md
        json.dump(tj, open(merge, "w"), indent=1)
        print(f"merged PMC figures of {sum(1 for k in per if k in tj['kernels'])} kernels into {merge}")


if __name__ == "__main__":
    main()
