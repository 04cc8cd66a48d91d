#!/bin/bash
# round 4: the MLP kernels of the unbounded-scene model's bf16 forward in cycles (GRBM / SQ counters, own --pmc pass): clock, MFMA-busy, parked waves.
# round 6: FORM=1 [default] = the one-kernel form, FORM=0 = k_pre_gemm + trunk; TAG names the output file
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_u16c
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python $ROOT/scripts/micro/prof_unbounded.py bf16 4 ${FORM:-1} > $OUT/pmc.log 2>&1
python - $OUT/pmc <<'PY' | tee -a $ROOT/gpurun_out/${TAG:-r06}_unbounded_bf16_cycles.txt
import csv, sys, glob, collections
d = sys.argv[1]
M = 8192 * 256
rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
trace = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
for key, name, mfma_per_wave_tile in (("k_pre_gemm", "k_pre_gemm", 672), ("v4pre", "trunk k_mlp_bf16", 1120), ("v4fused", "one-kernel form k_mlp_bf16", 1792),
                                      ("cast_ipe_360_tile", "k_cast_ipe_360_tile", 0)):
    acc = collections.defaultdict(list)
    for r in rows:
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace if key in r["Kernel_Name"])
    if not durs:
        continue
    g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
    du = durs[len(durs) // 2]
    w = {c: sum(v) / len(v) for c, v in acc.items()}
    mf = (M / 32) * mfma_per_wave_tile * 32 / 1024            # cycles a SIMD's matrix pipe is busy: 32 per v_mfma_f32_32x32x16_bf16
    print(f"{name}: median {du:.0f} us (under counters), cycles/XCD {g:.0f} -> clock {g / du / 1e3:.3f} GHz, MFMA-busy {mf / g:.3f} of the cycles, "
          f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES {w['SQ_WAIT_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}, "
          f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES {w['SQ_ACTIVE_INST_ANY'] / w['SQ_WAVE_CYCLES']:.3f}")
PY
rm -rf $OUT
