#!/bin/bash
# End-to-end rehearsal of the N > 1 launch path on a one-GPU box: every plugin's train.py through `python -m torch.distributed.run
# --nproc-per-node 1` (what scripts/train.sh does for --gpus a,b,...), i.e. env:// rendezvous, RCCL process group, ssecg.parallel.DataParallel +
# SyncBatchNorm, hip_graph auto (on: <= 128 windows over nccl, collectives captured), evaluate() with its gathers, checkpoints - then test.py on
# the checkpoint.  SSECG_FORCE_SYNC_COLLECTIVES=1: a one-rank group issues every collective of the N > 1 step.
# usage (GPU box): bash tools/e2e_torchrun_one_rank.sh [outdir]
OUT=${1:-gpurun_out/e2e_torchrun}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
CKPT=$(mktemp -d /tmp/e2e_torchrun.XXXXXX)   # checkpoints stay off gpurun_out/ (64 MiB are merged back), only the logs go there
cd "$(dirname "$0")/../semi-seg-ecg_amd" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 SSECG_FORCE_SYNC_COLLECTIVES=1
rc=0
port=23450
for cfg in fixmatch_smallbatch_synthetic mean_teacher_synthetic cps_synthetic stpp_synthetic base_synthetic; do
  port=$((port + 1))
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
      train.py --config_path configs/$cfg.yaml --output_dir "$CKPT/$cfg" --exp_name run > "$OUT/$cfg.log" 2>&1
  r=$?
  echo "$cfg: train.py exit $r | $(grep -c 'hip graph\|HIP graph' "$OUT/$cfg.log") graph lines | $(grep -E 'Training time|MeanIoU' "$OUT/$cfg.log" | tail -1 | cut -c1-160)"
  [ $r -ne 0 ] && { rc=1; tail -5 "$OUT/$cfg.log"; }
done
rm -rf "$CKPT"
exit $rc
