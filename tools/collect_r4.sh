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
{ echo "# bash tools/dist_overhead.sh (one MI355X, world-size-1 RCCL group, collectives forced; four steady-state steps of a rocprofv3 kernel trace)"
  echo "# round 3 (torch DDP, async SyncBN collectives everywhere): +1.13 ms/step.  Round 4, same protocol, in order of the changes:"
  echo "#   torch DDP as it was at the start of the round ......................................... +1.50 ms (65 mul_out launches, 42 async BN collectives)"
  echo "#   own reducer (ssecg/parallel.py: one staging launch per bucket, .grad inside the bucket)  +1.08 ms"
  echo "#   forward BN collectives synchronous on the current stream, downsample pairs merged ...... +0.84 ms (idle per BatchNorm 25 us -> 10 us)"
  echo "#   backward BN collectives synchronous where no weight gradient is pending to overlap ..... +0.55 ... +0.65 ms (boxes)"
  echo "# idle gaps that remain (tools/kernel_sequence.sh): 18 x 10 us forward, 10 x 10 us + 11 x 13 us backward, 5 x 14 us gradient buckets."
  echo "# final build:"
  grep -v amdgpu.ids $G/$TAG/dist_overhead.txt; } > profiles/r04_dist_overhead_one_rank.txt
{ echo "# python tools/amp_bench.py 1024 (back-to-back protocol: 10 launches between one event pair after a ~1 ms filler; this"
  echo "# matches rocprofv3 for compute-bound launches, but the operands of one layer stay warm in the Infinity Cache between"
  echo "# launches, which flatters the HBM-bound BatchNorm columns -- the in-step averages of r04_bench_amp_kernel_stats.md are"
  echo "# the ground truth for those)"
  grep -v amdgpu.ids $G/$TAG/amp_microbench.txt; } > profiles/r04_amp_microbench.txt
ls -la profiles | grep r04_
