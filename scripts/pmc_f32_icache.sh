#!/bin/bash
# round 4: instruction-cache behaviour of k_mlp_f32r (150 KB of straight-line code per tile through a 64-KB instruction cache)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ic
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -o pmc -- python $ROOT/bench.py --mode inference --precision fp32 --steps 4 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 > $OUT/pmc.log 2>&1
python - $OUT/pmc <<'PY' | tee $ROOT/gpurun_out/${F32_IC_OUT:-r04u_f32r_icache.txt}
import csv, sys, glob, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    if "k_mlp_f32" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = {c: sum(v) / len(v) for c, v in acc.items()}
print({k: round(v) for k, v in w.items()})
if "SQC_ICACHE_REQ" in w:
    print(f"icache hit rate {w['SQC_ICACHE_HITS'] / w['SQC_ICACHE_REQ']:.4f}, misses per launch {w['SQC_ICACHE_MISSES']:.0f} (+{w.get('SQC_ICACHE_MISSES_DUPLICATE', 0):.0f} duplicate), "
          f"requests {w['SQC_ICACHE_REQ']:.0f}; per 128-sample tile: {w['SQC_ICACHE_MISSES'] / 4096:.0f} misses of {w['SQC_ICACHE_REQ'] / 4096:.0f} requests")
PY
rm -rf $OUT/pmc
