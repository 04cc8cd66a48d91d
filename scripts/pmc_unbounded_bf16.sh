#!/bin/bash
# round 4: HBM bytes per launch of the kernels of the unbounded-scene model's bf16 forward, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (FETCH x2
# on gfx950, MI355X_MICROARCH.md), 8192 rays x (256 + 256) samples.  Round 6: the forward is k_cast_ipe_360_tile + ONE MLP kernel (the one-kernel form);
# FORM=0 measures the two-kernel form (k_pre_gemm + trunk).  The result is merged into gpurun_out/pmc_traffic.json ("unbounded_bf16") when that file exists.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_u16; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $ROOT/scripts/micro/prof_unbounded.py bf16 2 ${FORM:-1} > $OUT/$c.log 2>&1; echo "pmc $c rc=$?"
done
python - $OUT <<'PY' | tee $ROOT/gpurun_out/${TAG:-r05}_unbounded_bf16_traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
def means(counter):
    f = glob.glob(f"{out}/{counter}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        k = ("k_cast_ipe_360_tile" if "cast_ipe_360_tile" in n else "k_pre_gemm" if "k_pre_gemm" in n else "k_mlp_bf16 (one-kernel form)" if "fused" in n
             else "k_mlp_bf16 (trunk)" if "k_mlp_bf16" in n else None)
        if k:
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = means("FETCH_SIZE"), means("WRITE_SIZE")
M = 8192 * 256
res = {"samples_per_launch": M, "kernels": {}}
for k in fe:
    print(f"{k}: FETCH {fe[k] * 2 * 1024 / 1e9:.3f} GB (x2 corrected) + WRITE {wr.get(k, 0) * 1024 / 1e9:.3f} GB per launch of {M} samples")
    res["kernels"][k] = {"FETCH_SIZE_KB": round(fe[k], 1), "WRITE_SIZE_KB": round(wr.get(k, 0.0), 1), "hbm_bytes_per_launch": int((2 * fe[k] + wr.get(k, 0.0)) * 1024)}
import json, os
pj = os.path.join(os.path.dirname(out), "pmc_traffic.json")
if os.path.exists(pj):
    j = json.load(open(pj))
    j["unbounded_bf16"] = res
    json.dump(j, open(pj, "w"), indent=1)
    print("merged into", pj)
PY
rm -rf $OUT
