#!/bin/bash
# round 5: HBM bytes per launch of the kernels of the unbounded model's one-graph bf16 training step (4096 rays x (128 + 128) samples),
# FETCH_SIZE / WRITE_SIZE in separate --pmc passes (FETCH x2 on gfx950, MI355X_MICROARCH.md)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_t360; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $ROOT/scripts/micro/prof_train360.py bf16_graph > $OUT/$c.log 2>&1; echo "pmc $c rc=$?"
done
python - $OUT <<'PY' | tee $ROOT/gpurun_out/${TAG:-r05}_train360_traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
NAMES = (("k_mlp_wgrad", "k_mlp_wgrad"), ("trainfwd_pre", "k_mlp_bf16_trainfwd_pre"), ("k_mlp_bf16_dgrad", "k_mlp_bf16_dgrad"), ("k_pre_gemm", "k_pre_gemm"),
         ("cast_ipe_360_tile", "k_cast_ipe_360_tile"))
def means(counter):
    f = glob.glob(f"{out}/{counter}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        for pat, k in NAMES:
            if pat in r["Kernel_Name"]:
                acc[k].append(float(r["Counter_Value"]))
                break
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = means("FETCH_SIZE"), means("WRITE_SIZE")
tot = 0.0
for k in fe:
    f, w = fe[k] * 2 * 1024 / 1e9, wr.get(k, 0) * 1024 / 1e9
    tot += 2 * (f + w)
    print(f"{k}: FETCH {f:.3f} GB (x2 corrected) + WRITE {w:.3f} GB per launch (one level of 524288 samples)")
print(f"both levels, these kernels: {tot:.2f} GB per training step")
PY
rm -rf $OUT
