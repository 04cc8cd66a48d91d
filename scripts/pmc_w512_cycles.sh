#!/bin/bash
# round 5: the 512-wide trunk's bf16 kernel (one wave per SIMD) in cycles (GRBM / SQ counters, own --pmc pass): clock, MFMA-busy, parked waves;
# next to the shipped shape's kernel from the same script for comparison
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_w512
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in w512 default; do
  rm -rf $OUT/$v
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$v -o pmc -- python $ROOT/scripts/micro/time_variant_mlp.py $v > $OUT/$v.log 2>&1
done
python - $OUT <<'PY' | tee $ROOT/gpurun_out/${TAG:-r05}_w512_cycles.txt
import csv, sys, glob, collections
out = sys.argv[1]
M = 4096 * 128
for v, key, mfma_per_wave_tile in (("w512", "v610k_mlp_bf16", 4608), ("default", "3mip10k_mlp_bf16", 1216)):
    rows = list(csv.DictReader(open(glob.glob(f"{out}/{v}/**/*counter_collection.csv", recursive=True)[0])))
    trace = list(csv.DictReader(open(glob.glob(f"{out}/{v}/**/*kernel_trace.csv", recursive=True)[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace if key in r["Kernel_Name"])
    if not durs:
        print(v, "kernel not found", sorted({r["Kernel_Name"][:60] for r in trace})[:8])
        continue
    g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
    du = durs[len(durs) // 2]
    w = {c: sum(x) / len(x) for c, x in acc.items()}
    mf = (M / 32) * mfma_per_wave_tile * 32 / 1024            # cycles a SIMD's matrix pipe is busy: 32 per v_mfma_f32_32x32x16_bf16
    print(f"{v} bf16 kernel: median {du:.0f} us (under counters), cycles/XCD {g:.0f} -> clock {g / du / 1e3:.3f} GHz, MFMA-busy {mf / g:.3f} of the cycles, "
          f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, "
          f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES {w['SQ_ACTIVE_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}")
PY
rm -rf $OUT
