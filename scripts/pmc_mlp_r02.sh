#!/bin/bash
# Round-2 state of the headline kernel in cycles: SQ counters of k_mlp_bf16 (separate --pmc pass, kernel-trace only) + the same
# instruction stream on all-zero operands (nothing toggles: what the power envelope costs).   usage: pmc_mlp_r02.sh OUTDIR
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/pmc_mlp}; case $OUT in /*) ;; *) OUT=$ROOT/$OUT;; esac
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in random zero; do
  extra=""; [ $mode = zero ] && extra="--zero"
  for r in 1 2 3; do timeout 120 python $ROOT/scripts/prof_mlp.py --iters 20 $extra 2>/dev/null | sed "s/^/$mode: /"; done
  rm -rf $OUT/pmc_$mode
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$mode -o pmc -- python $ROOT/scripts/prof_mlp.py --iters 6 $extra > $OUT/pmc_$mode.log 2>&1
  python - $OUT/pmc_$mode $mode <<'PY'
import csv, sys, glob, collections
d, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    if "k_mlp" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])) if "k_mlp" in r["Kernel_Name"]]
print(f"{mode}: dur_us", ["%.0f" % x for x in durs])
for k, v in sorted(acc.items()):
    print(f"{mode}:   {k}: {sum(v)/len(v):.5g}")
g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
print(f"{mode}:   cycles/XCD {g:.0f}  -> clock {g / (sum(durs)/len(durs)) / 1e3:.3f} GHz ; MFMA-busy fraction {1216*8*2*32/g:.3f} (1216 chunks x 8 tiles x 2 waves/SIMD x 32 cycles)")
PY
  rm -rf $OUT/pmc_$mode
done 2>&1 | tee $OUT/pmc_mlp_r02.txt
