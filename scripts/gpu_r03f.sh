#!/bin/bash
# round 3: IPE-in-shadow A/B (libmipnerf_hip.so = shadow, libmipnerf_hip_noshadow.so = encoding phase at tile start)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_stages.py tests/test_gpu_reference_parity.py -m gpu -q -k "not trajectory and not converged and not full_size_training" 2>&1 | tail -15 > gpurun_out/r03f_pytest.txt
tail -4 gpurun_out/r03f_pytest.txt
LOG=gpurun_out/r03f_ab.log; : > $LOG
A=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_hip.so
B=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_hip_noshadow.so
for round in 1 2 3; do
  for so in $A $B; do MIPNERF_LIB=$so timeout 120 python scripts/prof_fwd.py --iters 200 --heat 1.0 >> $LOG 2>&1; done
done
cat $LOG
cd /tmp
for so in $A $B; do
  tag=$(basename $so .so)
  rm -rf /tmp/pmc_$tag
  MIPNERF_LIB=$so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/prof_fwd.py --iters 6 > /tmp/pmc_$tag.log 2>&1
  python - /tmp/pmc_$tag $tag <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r03f_pmc.txt
import csv, sys, glob, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    if "k_mlp_bf16" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])) if "k_mlp_bf16" in r["Kernel_Name"]]
print(f"{tag}: dur_us mean {sum(durs)/len(durs):.1f} (n={len(durs)})")
for k, v in sorted(acc.items()):
    print(f"{tag}:   {k}: {sum(v)/len(v):.5g}")
g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
print(f"{tag}:   cycles/XCD {g:.0f} -> clock {g / (sum(durs)/len(durs)) / 1e3:.3f} GHz ; MFMA-busy fraction {1216*8*2*32/g:.3f}")
PY
done
