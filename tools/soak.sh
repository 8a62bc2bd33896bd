#!/bin/bash
# Soak: 500 timed steps twice in a row for the fp32 line, the bf16 line and the 16-window HIP-graph line; prints ms/step and the final statistics
# of every run (pairs must agree bit for bit: every summation order on the path is fixed).  usage (GPU box): bash tools/soak.sh > gpurun_out/soak.txt
for cfg in "" "--amp" "--batch 16 --graph"; do
  for run in 1 2; do
    python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-amp-record $cfg 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s run $run  %7.3f ms/step  final %s' % ('[$cfg]', j['ms_per_step'], j['final_stats']))"
  done
done
