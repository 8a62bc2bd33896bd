#!/bin/bash
# Round-3 end-of-round evidence run on the GPU box: full GPU test suite, fp32 and bf16 bench lines (20 + 100 steps), rocprofv3
# kernel stats + FETCH/WRITE PMC passes of both lines, small-batch (BASELINE config #2 shape) lines and launch census.
# usage: bash tools/e2e_r3.sh <tag>
TAG=${1:-r3z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/tests_all.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests_all.log; tail -2 $OUT/tests_all.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err && echo bench fp32 done
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_fp32_100.json 2>> $OUT/bench_fp32.err
python bench.py --steps 20 --warmup 5 --amp --no-cpu-baseline > $OUT/bench_amp.json 2> $OUT/bench_amp.err
python bench.py --steps 100 --warmup 20 --amp --no-cpu-baseline > $OUT/bench_amp_100.json 2>> $OUT/bench_amp.err
python bench.py --steps 20 --warmup 5 --batch 256 --leads 1 --no-cpu-baseline > $OUT/bench_b256c1_fp32.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --batch 256 --leads 1 --no-cpu-baseline --amp > $OUT/bench_b256c1_amp.json 2> /dev/null
bash tools/profile_bench.sh ${TAG}_fp32 > $OUT/profile_fp32.log 2>&1
bash tools/profile_bench.sh ${TAG}_amp --amp > $OUT/profile_amp.log 2>&1
bash tools/launch_census.sh $TAG/census_b256c1_fp32 --batch 256 --leads 1 > $OUT/census_b256c1_fp32.txt 2>&1
python tools/host_profile.py 256 > $OUT/host_b256.txt 2>&1
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], j["dtype"], round(j["ms_per_step"], 3), "ms/step", round(j["value"]), "windows/s", j["roofline"]["kernel"], round(j["roofline"]["frac"], 3))
PY
done
