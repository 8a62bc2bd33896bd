#!/bin/bash
# Round-6 end-of-round evidence run on the GPU box (kernel sources frozen: every summary is keyed by their hash).
#   part A (default): default bench line (fp32 headline + bf16 `amp` sub-record + CPU baseline), 100-step lines, rocprofv3 kernel
#                     stats + FETCH / WRITE passes of both precisions, SQ counter passes of the fp32 convolution kernels
#   part B (B):       small-batch eager / HIP-graph bench, one-rank RCCL overhead + the `dist` record, use_amp parity printout
# usage: bash tools/e2e_r6.sh <tag> [A|B]
TAG=${1:-r6z}
PART=${2:-A}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$PART" = A ]; then
  python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "default bench done"
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-amp-record > $OUT/bench_fp32_100.json 2> $OUT/bench_fp32_100.err
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --amp > $OUT/bench_amp_100.json 2> $OUT/bench_amp_100.err
  bash tools/profile_bench.sh ${TAG}_fp32 > $OUT/profile_fp32.log 2>&1 && echo "fp32 profile done"
  bash tools/profile_bench.sh ${TAG}_amp --amp > $OUT/profile_amp.log 2>&1 && echo "bf16 profile done"
  bash tools/pmc_kernel.sh conv_ -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amp-record > $OUT/pmc_conv_fp32.txt 2>&1 && echo "fp32 SQ counters done"
  bash tools/pmc_kernel.sh b16 -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --amp > $OUT/pmc_amp_b16.txt 2>&1 && echo "bf16 SQ counters done"
else
  CFGS="256 1;64 12;16 12" bash tools/graph_bench.sh $TAG/graph > $OUT/graph_bench.txt 2>&1 && echo "small-batch bench done"
  bash tools/dist_overhead.sh gpurun_out/$TAG/dist > $OUT/dist_overhead.txt 2>&1
  python tools/wino_wgrad_bench.py > $OUT/wino_wgrad_bench.txt 2>&1
  SSECG_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-amp-record > $OUT/bench_one_rank_rccl.json 2> $OUT/bench_one_rank_rccl.err
  bash tools/graph_bench_dist.sh $TAG/graph_dist > $OUT/graph_bench_dist.txt 2>&1 && echo "forced-dist graph bench done"
  python -m pytest tests/test_ampfix_gpu.py -q -s 2>&1 | grep -E "^\.?(layer|head|stem|ampfix_|autograd chain|loss_x|[0-9]+ passed)" > $OUT/ampfix_gpu.txt
  tail -3 $OUT/ampfix_gpu.txt
fi
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], j["dtype"], round(j["ms_per_step"], 3), "ms/step", round(j["value"]), "windows/s", j["roofline"]["kernel"], round(j["roofline"]["frac"], 3),
      ("| amp sub-record " + str(round(j["amp"]["ms_per_step"], 3)) + " ms/step") if "amp" in j else "", ("| dist " + json.dumps(j["dist"]["collectives_per_step"])) if "dist" in j else "")
PY
done
