#!/bin/bash
# cycles (not wall time) per MLP dispatch for each variant .so: separates cycle efficiency from DVFS clock
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for so in $ROOT/mipnerf_pl_amd/csrc/libmipnerf_hip*.so; do
  tag=$(basename $so .so)
  rm -rf $OUT/abpmc_$tag
  MIPNERF_LIB=$so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/abpmc_$tag -o pmc -- python $ROOT/scripts/prof_mlp.py --iters 6 ${PROF_ARGS:-} > $OUT/abpmc_$tag.log 2>&1
  echo "== $tag rc=$?"
  python - $OUT/abpmc_$tag <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
    if "k_mlp" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])) if "k_mlp" in r["Kernel_Name"]]
print("  dur_us", ["%.0f" % x for x in durs])
for k, v in sorted(acc.items()):
    print(f"  {k}: {sum(v)/len(v):.5g}")
g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
print(f"  cycles/XCD {g:.0f}  -> clock {g / (sum(durs)/len(durs)) / 1e3:.3f} GHz ; MFMA-busy frac {1216*8*2*32/g:.3f}")
PY
done
