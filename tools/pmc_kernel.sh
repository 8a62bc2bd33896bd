#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, --kernel-trace only) over the kernels whose name contains $1, for the
# command given after "--" (a python script path relative to the repo + its arguments).
# usage (GPU box): bash tools/pmc_kernel.sh stem_ -- tools/stem_bench.py 1024
PAT=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$PAT
mkdir -p $OUT
i=0
IFS=";" read -ra GROUPS_ <<< "${PMC_GROUPS:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS;SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY}"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python3 $GRAFT_REPO_ROOT/$1 "${@:2}" > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "  (no counter file; see $OUT/g$i.log)"; grep -m2 -i "error\|exceeds" $OUT/g$i.log; continue; fi
  python3 - "$f" "$PAT" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
        a = acc[(k, r["Grid_Size"], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, g, c), (v, n) in sorted(acc.items()):
    print(f"{k:34s} grid {g:>8s} {c:28s} {v / n:16.0f} per launch ({n})")
PY
  rm -rf $OUT/g$i
done
