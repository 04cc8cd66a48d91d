# HBM traffic of the three bf16 training kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the eager
# training bench (summary -> profiles/r02_pmc_train.json).   usage: [OUT=dir] pmc_train.sh
export TMPDIR=/tmp
OUT=${OUT:-$GRAFT_REPO_ROOT/gpurun_out/pmc_train}; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY' | tee -a $OUT/pmc_train_summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "wgrad" if "k_mlp_wgrad" in n else "trainfwd" if "trainfwd" in n else "dgrad" if "dgrad" in n else None
    if k: acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[2]} {k}: mean per dispatch {sum(v)/len(v):.6g} KB (n={len(v)})")
PY
  rm -rf $OUT/pmc_$c
done
