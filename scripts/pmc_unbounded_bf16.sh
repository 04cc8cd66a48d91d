#!/bin/bash
# round 4: HBM bytes per launch of the three kernels of the unbounded-scene model's bf16 forward (k_cast_ipe_360_frag, k_pre_gemm, trunk k_mlp_bf16),
# FETCH_SIZE / WRITE_SIZE in separate --pmc passes (FETCH x2 on gfx950, MI355X_MICROARCH.md), 8192 rays x (256 + 256) samples
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_u16; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $ROOT/scripts/micro/prof_unbounded.py bf16 2 > $OUT/$c.log 2>&1; echo "pmc $c rc=$?"
done
python - $OUT <<'PY' | tee $ROOT/gpurun_out/${TAG:-r05}_unbounded_bf16_traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
def means(counter):
    f = glob.glob(f"{out}/{counter}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        k = "k_cast_ipe_360_tile" if "cast_ipe_360_tile" in n else "k_pre_gemm" if "k_pre_gemm" in n else "k_mlp_bf16 (trunk)" if "k_mlp_bf16" in n else None
        if k:
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = means("FETCH_SIZE"), means("WRITE_SIZE")
M = 8192 * 256
for k in fe:
    print(f"{k}: FETCH {fe[k] * 2 * 1024 / 1e9:.3f} GB (x2 corrected) + WRITE {wr.get(k, 0) * 1024 / 1e9:.3f} GB per launch of {M} samples")
PY
rm -rf $OUT
