#!/bin/bash
# round 4: when the box runs k_mlp_f32r slowly (launch > 18.5 ms at configs[3]), say why: SQ cycle counters, L2 hit rate, LDS stalls,
# the go / no-go micro-kernel (no HBM traffic at all), the in-process MFMA ceilings and rocm-smi on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=$(timeout 200 python bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print(l['roofline']['launch_ms'])")
echo "launch_ms $L"
if python -c "import sys; sys.exit(0 if float('$L') > 18.5 else 1)" || [ "${FORCE_PROBE:-0}" = "1" ]; then
  echo "PROBED BOX (slow if launch_ms > 18.5)"
  F32_CYCLES_OUT=r04s_slow_box_cycles.txt bash scripts/pmc_f32_cycles.sh 2>&1 | tail -1
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sb; mkdir -p $OUT; cd /tmp
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    rm -rf $OUT/p
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode inference --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 > $OUT/pmc.log 2>&1
    python - $OUT/p <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0])):
    if "k_mlp_f32" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
  done
  rm -rf $OUT
  cd $GRAFT_REPO_ROOT
  timeout 60 scripts/micro/f32r_probe.out 1.0
  python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from mipnerf_pl_amd import _lib as L
D = L.diag_lib()
for mode, nm in ((10, "fp32 register-fed"), (0, "bf16 register-fed"), (2, "bf16 lds+dma-fed")):
    r = (C.c_double * 3)()
    L.diag_check(D.mipnerf_mfma_ceiling(mode, 2, 1, 1.2, r, torch.cuda.current_stream().cuda_stream), "c")
    print(nm, "ceiling", round(r[0], 1), "TF", round(r[2], 3), "GHz")
PY
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "clock|Power" | head -8
fi
