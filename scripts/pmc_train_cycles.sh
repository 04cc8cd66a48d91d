#!/bin/bash
# round 3: the three bf16 training kernels in cycles (separate --pmc pass, kernel-trace only): GRBM cycles per XCD -> effective clock and
# the MFMA-busy fraction implied by the executed MFMA count (algorithmic flops x 1.02 padding x 1.11 transposition MFMAs for the two
# producers; wgrad: algorithmic + the bias tile).  usage: pmc_train_cycles.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_train_cycles
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python $ROOT/scripts/prof_train.py --iters 8 > $OUT/pmc.log 2>&1
python - $OUT/pmc <<'PY' | tee $ROOT/gpurun_out/r03ae_train_cycles.txt
import csv, sys, glob, collections
d = sys.argv[1]
M = 524288
# executed MFMA flops per launch (one level, 524,288 samples)
flops = {"trainfwd": 1220608 * M * 1.02 * 1.11, "dgrad": 1115392 * M * 1.02 * 1.11, "k_mlp_wgrad": 1220608 * M * (1 + 1 / 8.0) * 1.0}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    for k in flops:
        if k in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    for k in flops:
        if k in r["Kernel_Name"]:
            durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in flops:
    if not acc[k]:
        continue
    g = sum(acc[k]["GRBM_GUI_ACTIVE"]) / len(acc[k]["GRBM_GUI_ACTIVE"]) / 8
    du = sorted(durs[k])[len(durs[k]) // 2]
    mf = flops[k] / 32768 / 1024 * 32          # MFMA cycles per SIMD
    w = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print(f"{k}: median {du:.0f} us, cycles/XCD {g:.0f} -> clock {g / du / 1e3:.3f} GHz, MFMA-busy {mf / g:.3f} of the cycles, "
          f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {w.get('SQ_WAIT_ANY', 0) / max(w.get('SQ_WAVE_CYCLES', 1), 1):.3f}, "
          f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES {w.get('SQ_ACTIVE_INST_ANY', 0) / max(w.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
rm -rf $OUT/pmc
