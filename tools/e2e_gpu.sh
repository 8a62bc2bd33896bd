#!/bin/bash
# End-to-end check of the algorithm plugins through train.py and test() on generated windows.
set -e
cd "$(dirname "$0")/../semi-seg-ecg_amd"
for cfg in base_synthetic mean_teacher_synthetic cps_synthetic stpp_synthetic fixmatch_devaug_synthetic; do
  echo "== $cfg"
  timeout -k 10 600 python train.py --config_path configs/$cfg.yaml --output_dir /tmp/exps_$cfg 2>&1 | grep -E "Averaged stats|MeanIoU:|Training time|Load teacher|Error|Traceback" | tail -14
done
python - <<'PY'
import sys, yaml
sys.path.insert(0, ".")
import algorithms
cfg = yaml.safe_load(open("configs/base_synthetic.yaml")); cfg["output_dir"] = "/tmp/exps_base_synthetic"
print("test():", algorithms.base.test(cfg))
PY
