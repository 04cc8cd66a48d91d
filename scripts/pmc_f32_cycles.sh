#!/bin/bash
# rounds 3-4: the fp32 MLP kernel (k_mlp_f32 in round 3, the register-resident k_mlp_f32r since round 4) in cycles: GRBM / SQ counters of the headline-shaped fp32 forward (4096 x (128+128))
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_f32
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python $ROOT/bench.py --mode inference --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 > $OUT/pmc.log 2>&1
python - $OUT/pmc <<'PY' | tee $ROOT/gpurun_out/${F32_CYCLES_OUT:-r04e_f32r_cycles.txt}
import csv, sys, glob, collections
d = sys.argv[1]
M = 524288
flop = 1220608 * M      # k_mlp_f32r executes exactly the algorithmic MACs on the matrix pipe (k_mlp_f32 of round 3: x 1.02)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    if "k_mlp_f32" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])) if "k_mlp_f32" in r["Kernel_Name"]]
g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
du = sorted(durs)[len(durs) // 2]
mf = flop / 4096 / 1024 * 64
w = {c: sum(v) / len(v) for c, v in acc.items()}
print(f"k_mlp_f32: median {du:.0f} us, cycles/XCD {g:.0f} -> clock {g / du / 1e3:.3f} GHz, MFMA-busy {mf / g:.3f} of the cycles, "
      f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, "
      f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES {w['SQ_ACTIVE_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, SQ_BUSY_CYCLES / GRBM {w['SQ_BUSY_CYCLES'] / sum(acc['GRBM_GUI_ACTIVE']) * len(acc['GRBM_GUI_ACTIVE']):.3f}")
PY
rm -rf $OUT/pmc
