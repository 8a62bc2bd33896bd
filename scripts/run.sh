#!/bin/bash
# Launcher with the reference's command line (scripts/train.sh, test.sh, inferernce.sh of bakqui/semi-seg-ecg: same option names),
# for the MI355X source root semi-seg-ecg_amd/.  One process per GPU; more than one GPU goes through torch.distributed.run with
# RCCL (backend "nccl") over xGMI.  Invoked through the three thin wrappers next to it:
#   bash scripts/train.sh --gpus 0,1,2,3 -f configs/fixmatch_synthetic.yaml [-o override.yaml] [--output_dir D] [--exp_name N]
#                         [--resume ckpt.pth] [--start_epoch E] [--master_port P]
#   bash scripts/test.sh / scripts/inference.sh --gpus 0 -f cfg.yaml [-o override.yaml] [--model_path ckpt.pth] [--output_dir D]
ENTRY=$1; shift
PORT=12345; GPUS=0; ARGS=()
while [ $# -gt 0 ]; do
  case "$1" in
    --master_port) PORT=$2; shift 2 ;;
    --gpus) GPUS=$2; shift 2 ;;
    -f|--config_path) ARGS+=(--config_path "$2"); HAVE_CFG=1; shift 2 ;;
    -o|--override_config_path) ARGS+=(--override_config_path "$2"); shift 2 ;;
    --output_dir|--exp_name|--resume|--start_epoch|--model_path) ARGS+=("$1" "$2"); shift 2 ;;
    -h|--help) sed -n 2,9p "$0"; exit 0 ;;
    *) echo "unknown option: $1" >&2; exit 2 ;;
  esac
done
[ -n "${HAVE_CFG:-}" ] || { echo "error: -f / --config_path is required" >&2; exit 2; }
export HIP_VISIBLE_DEVICES=$GPUS                 # the reference sets CUDA_VISIBLE_DEVICES
export HSA_ENABLE_IPC_MODE_LEGACY=0             # RCCL / device-tensor sharing across processes needs dmabuf IPC on this stack
N=$(echo "$GPUS" | tr ',' '\n' | grep -c .)
cd "$(dirname "$0")/../semi-seg-ecg_amd" || exit 1
if [ "$N" -gt 1 ]; then
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" "$ENTRY" "${ARGS[@]}"
else
  exec python "$ENTRY" "${ARGS[@]}"
fi
