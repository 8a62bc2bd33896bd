#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: kernel stats (one pass) + FETCH_SIZE / WRITE_SIZE PMC (separate passes,
# no trace domains combined with --pmc beyond --kernel-trace).  Outputs under gpurun_out/prof_<tag>/; summarise with
# tools/summarize_profile.py and copy the summary into profiles/.
# usage: bash tools/profile_bench.sh <tag> [extra bench.py arguments, e.g. --amp]
set -u
TAG=${1:-run}
[ $# -gt 0 ] && shift
EXTRA="$@"
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# Per-kernel durations are taken with the step on ONE stream (SSECG_OVERLAP_PASSES=0): in the default step the pseudo-label pass runs
# on a side stream beside the student forward, launches of the two passes share the chip and their durations are not per-kernel
# quantities (bench.py's instrumented step, which the roofline record is taken from, is single-stream for the same reason).  The
# trace of the default two-stream step is kept beside it (kernel_stats_two_streams.csv).
echo "kernel stats pass (single stream)"
SSECG_OVERLAP_PASSES=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-amp-record $EXTRA > $OUT/stats.log 2>&1
echo "kernel stats pass (default: two streams)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o r -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-amp-record $EXTRA > $OUT/stats2.log 2>&1
cp $(find $OUT/stats2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_two_streams.csv
rm -rf $OUT/stats2
echo "FETCH_SIZE pass"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o r -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amp-record $EXTRA > $OUT/fetch.log 2>&1
echo "WRITE_SIZE pass"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o r -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-amp-record $EXTRA > $OUT/write.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/fetch.csv
cp $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/write.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write
tail -1 $OUT/stats.log | cut -c1-300
ls -la $OUT
