#!/bin/bash
# PMC passes for the MLP kernel (separate from kernel-trace runs, as the guide prescribes).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
[ -f $OUT/counters.txt ] || rocprofv3 -L > $OUT/counters.txt 2>&1
python $ROOT/scripts/prof_mlp.py --iters 20 | tee $OUT/mlp_time_$TAG.log
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${TAG}_$i -o pmc -- python $ROOT/scripts/prof_mlp.py --iters 3 > $OUT/pmc_${TAG}_$i.log 2>&1
  echo "pmc set $i rc=$?"
  f=$(find $OUT/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "k_mlp" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k}: mean per dispatch {sum(v)/len(v):.6g} (n={len(v)})")
PY
done
