#!/bin/bash
# Copy what tools/e2e_r4.sh left under gpurun_out/<tag> into profiles/r04_* (run here, after the gpurun call returned).
# usage: bash tools/collect_r4.sh <tag>
set -eu
TAG=${1:-r4z}
G=gpurun_out
python tools/summarize_profile.py $G/prof_${TAG}_fp32/kernel_stats.csv $G/prof_${TAG}_fp32/fetch.csv $G/prof_${TAG}_fp32/write.csv 8 \
  profiles/r04_bench_kernel_stats.md "Round 4: FixMatch step kernel statistics, fp32 (final build)" 512 12 2000 f32
python tools/summarize_profile.py $G/prof_${TAG}_amp/kernel_stats.csv $G/prof_${TAG}_amp/fetch.csv $G/prof_${TAG}_amp/write.csv 8 \
  profiles/r04_bench_amp_kernel_stats.md "Round 4: FixMatch step kernel statistics, bf16 student pass (final build)" 512 12 2000 bf16
cp $G/prof_${TAG}_fp32/kernel_stats.csv profiles/r04_bench_kernel_stats.csv
cp $G/prof_${TAG}_amp/kernel_stats.csv profiles/r04_bench_amp_kernel_stats.csv
tail -1 $G/$TAG/bench_default.json > profiles/r04_bench_default.json
tail -1 $G/$TAG/bench_fp32_100.json > profiles/r04_bench_fp32_100.json
tail -1 $G/$TAG/bench_amp_100.json > profiles/r04_bench_amp_100.json
cp $G/$TAG/dist_overhead.txt profiles/r04_dist_overhead_one_rank.txt
{ echo "# python tools/amp_bench.py 1024 (back-to-back protocol: 10 launches between one event pair after a ~1 ms filler; this"
  echo "# matches rocprofv3 for compute-bound launches, but the operands of one layer stay warm in the Infinity Cache between"
  echo "# launches, which flatters the HBM-bound BatchNorm columns -- the in-step averages of r04_bench_amp_kernel_stats.md are"
  echo "# the ground truth for those)"
  grep -v amdgpu.ids $G/$TAG/amp_microbench.txt; } > profiles/r04_amp_microbench.txt
ls -la profiles | grep r04_
