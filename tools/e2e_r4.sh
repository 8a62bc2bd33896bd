#!/bin/bash
# Round-4 end-of-round evidence run on the GPU box (kernels frozen): the default bench line (fp32 headline + bf16 `amp` sub-record,
# CPU baseline), 100-step lines of both precisions, rocprofv3 kernel stats + FETCH / WRITE PMC passes of both, the one-rank RCCL
# overhead, the bf16 per-shape microbench.  usage: bash tools/e2e_r4.sh <tag>
TAG=${1:-r4z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "default bench done"
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-amp-record > $OUT/bench_fp32_100.json 2> $OUT/bench_fp32_100.err
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --amp > $OUT/bench_amp_100.json 2> $OUT/bench_amp_100.err
bash tools/profile_bench.sh ${TAG}_fp32 > $OUT/profile_fp32.log 2>&1
bash tools/profile_bench.sh ${TAG}_amp --amp > $OUT/profile_amp.log 2>&1
bash tools/dist_overhead.sh gpurun_out/$TAG/dist > $OUT/dist_overhead.txt 2>&1
python tools/amp_bench.py 1024 > $OUT/amp_microbench.txt 2>&1
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], j["dtype"], round(j["ms_per_step"], 3), "ms/step", round(j["value"]), "windows/s", j["roofline"]["kernel"], round(j["roofline"]["frac"], 3),
      ("| amp sub-record " + str(round(j["amp"]["ms_per_step"], 3)) + " ms/step") if "amp" in j else "")
PY
done
