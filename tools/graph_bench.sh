#!/bin/bash
# Eager vs HIP-graph replay of the whole step (bench.py --graph), same box: the headline shape and the small-batch shapes
# of BASELINE configs #2 / #4 (256 windows per GPU), fp32 and bf16.  usage: [CFGS="16 1;16 12"] bash tools/graph_bench.sh <tag>
OUT=gpurun_out/${1:-graph}
mkdir -p $OUT
IFS=";" read -ra CFGS_ <<< "${CFGS:-512 12;256 1;256 12;64 12}"
for cfg in "${CFGS_[@]}"; do
  set -- $cfg
  for amp in "" "--amp"; do
    for g in "" "--graph"; do
      name=b$1c$2${amp:+_amp}${g:+_graph}
      python bench.py --steps 40 --warmup 6 --batch $1 --leads $2 --no-cpu-baseline --no-amp-record $amp $g > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; exit 1; }
      python - $OUT/$name.json $name <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {j['ms_per_step']:8.3f} ms/step  {j['value']:9.0f} windows/s  device {j['device_ms_per_step']:.3f} ms", flush=True)
PY
    done
  done
done
