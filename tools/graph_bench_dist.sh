#!/bin/bash
# Eager vs whole-step HIP-graph replay UNDER torch.distributed on one GPU: a world-size-1 RCCL group with the data-parallel wrapper and the
# SyncBatchNorm all-reduces forced (SSECG_BENCH_FORCE_DIST=1) - every collective of the N > 1 step goes through ProcessGroupNCCL, eagerly
# (host-issued) and captured into the graph (round 6).  The reference's shipped operating point is 16 windows per GPU on several GPUs
# (configs/base/resnet18/fixmatch.yaml:86, scripts/train.sh:108-141).  usage: [CFGS="16 12;64 12"] bash tools/graph_bench_dist.sh <tag>
OUT=gpurun_out/${1:-graph_dist}
mkdir -p $OUT
export SSECG_BENCH_FORCE_DIST=1
IFS=";" read -ra CFGS_ <<< "${CFGS:-16 12;64 12}"
for cfg in "${CFGS_[@]}"; do
  set -- $cfg
  for amp in "" "--amp"; do
    for g in "" "--graph"; do
      name=dist_b$1c$2${amp:+_amp}${g:+_graph}
      python bench.py --steps 40 --warmup 6 --batch $1 --leads $2 --no-cpu-baseline --no-amp-record $amp $g > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; exit 1; }
      python - $OUT/$name.json $name <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = j["dist"]["collectives_per_step"]
print(f"{sys.argv[2]:30s} {j['ms_per_step']:8.3f} ms/step  {j['value']:9.0f} windows/s  device {j['device_ms_per_step']:.3f} ms  host {j['host_ms_per_step']:.3f} ms  "
      f"collectives/step {sum(v['count'] for v in c.values())}  final loss {j['final_stats']['loss_total']:.9g}  graph: {j['config']['hip_graph']}", flush=True)
PY
    done
  done
done
