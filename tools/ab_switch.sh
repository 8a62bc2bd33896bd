#!/bin/bash
# Same-box A/B of run-time switches on the headline bench (fp32 and bf16 lines, back to back, ABAB): usage
#   bash tools/ab_switch.sh <outdir> "SSECG_FUSE_BN=0" ["SSECG_BN_ROWS=0" ...]
OUT=gpurun_out/${1:-ab}; shift
mkdir -p $OUT
run() {   # name, env assignment (may be empty), extra bench flags
  env $2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-amp-record $3 > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 FAILED"; tail -3 $OUT/$1.err; return; }
  python - $OUT/$1.json "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:44s} {j['ms_per_step']:8.3f} ms/step  switches {j['config'].get('switches')}", flush=True)
PY
}
for rep in 1 2; do
  run base_fp32_$rep "" ""
  for sw in "$@"; do run "${sw//=/_}_fp32_$rep" "$sw" ""; done
done
for rep in 1 2; do
  run base_amp_$rep "" "--amp"
  for sw in "$@"; do run "${sw//=/_}_amp_$rep" "$sw" "--amp"; done
done
