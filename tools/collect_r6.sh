#!/bin/bash
# Copy what tools/e2e_r6.sh left under gpurun_out/<tag> into profiles/r06_* (run in the build container, after the gpurun calls returned;
# the kernel sources must be the ones the run measured: summarize_*.py stamp the summaries with their hash).
# usage: bash tools/collect_r6.sh <tag>
set -eu
TAG=${1:-r6z}
G=gpurun_out
if [ -f $G/prof_${TAG}_fp32/kernel_stats.csv ]; then
  python tools/summarize_profile.py $G/prof_${TAG}_fp32/kernel_stats.csv $G/prof_${TAG}_fp32/fetch.csv $G/prof_${TAG}_fp32/write.csv 8 \
    profiles/r06_bench_kernel_stats.md "Round 6: FixMatch step kernel statistics, fp32 (final build)" 512 12 2000 f32 > /dev/null
  python tools/summarize_profile.py $G/prof_${TAG}_amp/kernel_stats.csv $G/prof_${TAG}_amp/fetch.csv $G/prof_${TAG}_amp/write.csv 8 \
    profiles/r06_bench_amp_kernel_stats.md "Round 6: FixMatch step kernel statistics, bf16 student pass (final build)" 512 12 2000 bf16 > /dev/null
  cp $G/prof_${TAG}_fp32/kernel_stats.csv profiles/r06_bench_kernel_stats.csv
  cp $G/prof_${TAG}_amp/kernel_stats.csv profiles/r06_bench_amp_kernel_stats.csv
  cp $G/prof_${TAG}_fp32/kernel_stats_two_streams.csv profiles/r06_bench_kernel_stats_two_streams.csv
  cp $G/prof_${TAG}_amp/kernel_stats_two_streams.csv profiles/r06_bench_amp_kernel_stats_two_streams.csv
  python tools/summarize_pmc.py $G/$TAG/pmc_conv_fp32.txt profiles/r06_pmc_conv_fp32.md \
    "PMC counters of the fp32 convolution kernels in one FixMatch step (round 6, final build)" profiles/r06_bench_kernel_stats_traffic.json
  python tools/summarize_pmc.py $G/$TAG/pmc_amp_b16.txt profiles/r06_pmc_conv_bf16.md \
    "PMC counters of the bf16 convolution / weight-gradient / BatchNorm kernels in one FixMatch step under use_amp (round 6, final build; pattern b16, command bench.py --amp)" > /dev/null
  { echo "# bash tools/pmc_kernel.sh b16 -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --amp (round 6, final build; four rocprofv3 --pmc passes, per-launch averages)"
    grep -v amdgpu.ids $G/$TAG/pmc_amp_b16.txt; } > profiles/r06_amp_pmc_b16.txt
  tail -1 $G/$TAG/bench_default.json > profiles/r06_bench_default.json
  tail -1 $G/$TAG/bench_fp32_100.json > profiles/r06_bench_fp32_100.json
  tail -1 $G/$TAG/bench_amp_100.json > profiles/r06_bench_amp_100.json
fi
if [ -f $G/$TAG/graph_bench.txt ]; then
  { echo "Eager vs whole-step HIP-graph replay at small batches (CFGS=\"256 1;64 12;16 12\" bash tools/graph_bench.sh; b256c1 = BASELINE config #2's per-GPU shape; bench.py [--amp] [--graph] --steps 40 --warmup 6"
    echo "--no-cpu-baseline --no-amp-record --batch B --leads C), one MI355X, back to back, final build of round 6 (K split of small launches on by default)."
    echo "name = b<windows per GPU>c<leads>[_amp][_graph].  Round 5 on the same protocol: profiles/r05_graph_bench.txt."
    echo
    grep -v amdgpu.ids $G/$TAG/graph_bench.txt; } > profiles/r06_graph_bench.txt
  { echo "# bash tools/dist_overhead.sh (one MI355X, world-size-1 RCCL group, collectives forced; four steady-state steps of a rocprofv3 kernel trace), round 6 final build"
    grep -v amdgpu.ids $G/$TAG/dist_overhead.txt; } > profiles/r06_dist_overhead_one_rank.txt
  tail -1 $G/$TAG/bench_one_rank_rccl.json > profiles/r06_bench_one_rank_rccl.json
  { echo "# python tools/wino_wgrad_bench.py (one MI355X, N = 1024 windows, back-to-back launches between one event pair): the fp32 weight gradient of the"
    echo "# three-tap stride-1 convolutions as the transpose of F(2,3) (rounds 2-4; the 64-channel layer: the direct kernel) and of F(4,3) (round 5, default)"
    grep -v amdgpu.ids $G/$TAG/wino_wgrad_bench.txt; } > profiles/r06_wino_wgrad_bench.txt
  if [ -f $G/$TAG/graph_bench_dist.txt ]; then
    { sed -n 1,6p profiles/r06_graph_bench_dist.txt; grep -v amdgpu.ids $G/$TAG/graph_bench_dist.txt; } > profiles/r06_graph_bench_dist.tmp && mv profiles/r06_graph_bench_dist.tmp profiles/r06_graph_bench_dist.txt
  fi
  { echo "# tests/test_ampfix_gpu.py -s on one MI355X (round 6, final build): the HIP use_amp path against the reference executed under PyTorch's CPU bf16 autocast"
    cat $G/$TAG/ampfix_gpu.txt; } > profiles/r06_ampfix_gpu.txt
fi
ls -la profiles | grep r06_
