#!/bin/bash
# HBM bytes per launch of the MFMA kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# in SEPARATE passes, --kernel-trace only, FETCH_SIZE doubled on gfx950.  Writes gpurun_out/pmc_traffic.json (copy it to
# profiles/mlp_pmc.json: bench.py reads roofline.traffic from there and refuses a file of another round).
#   usage: ROUND=5 bash scripts/pmc_traffic.sh
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic; mkdir -p $OUT
ROUND=${ROUND:-6}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/inf_$c -o pmc -- python $ROOT/bench.py --mode inference --no-graph --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 --ceiling-seconds 0 > $OUT/inf_$c.log 2>&1; echo "pmc inference $c rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/train_$c -o pmc -- python $ROOT/bench.py --mode train --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-lightning-route --preheat-seconds 0 > $OUT/train_$c.log 2>&1; echo "pmc train $c rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/f32_$c -o pmc -- python $ROOT/bench.py --mode inference --no-graph --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 > $OUT/f32_$c.log 2>&1; echo "pmc fp32 $c rc=$?"
done
python - $OUT $ROUND <<'PY' | tee $ROOT/gpurun_out/pmc_traffic_summary.txt
import csv, glob, json, sys, collections
out, rnd = sys.argv[1], int(sys.argv[2])
def means(tag, counter, classify):
    f = glob.glob(f"{out}/{tag}_{counter}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            k = classify(r["Kernel_Name"])
            if k:
                acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
inf = lambda n: "k_mlp_bf16" if ("k_mlp_bf16" in n and "trainfwd" not in n and "dgrad" not in n) else None
trn = lambda n: "wgrad" if "k_mlp_wgrad" in n else "trainfwd" if "trainfwd" in n else "dgrad" if "dgrad" in n else None
f32 = lambda n: "k_mlp_f32r" if "k_mlp_f32r" in n else None
res = {"round": rnd, "gfx950_fetch_correction": 2.0,
       "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --mode inference / --mode train --no-graph / --precision fp32, scripts/pmc_traffic.sh, round {rnd}",
       "precision": "bf16", "samples_per_launch": 524288, "kernels": {}}
for tag, cl in (("inf", inf), ("train", trn), ("f32", f32)):
    fe, nf = means(tag, "FETCH_SIZE", cl)
    wr, nw = means(tag, "WRITE_SIZE", cl)
    for k in fe:
        b = int(fe[k] * 1024 * 2.0 + wr.get(k, 0.0) * 1024)
        res["kernels"][k] = {"FETCH_SIZE_KB": round(fe[k], 2), "WRITE_SIZE_KB": round(wr.get(k, 0.0), 2), "hbm_bytes_per_launch": b, "dispatches": nf[k]}
        print(f"{k}: FETCH {fe[k]:.6g} KB x2 + WRITE {wr.get(k, 0):.6g} KB = {b / 1e6:.2f} MB per launch (n = {nf[k]})")
k = res["kernels"]
if "k_mlp_bf16" in k:
    res.update({"FETCH_SIZE_KB": k["k_mlp_bf16"]["FETCH_SIZE_KB"], "WRITE_SIZE_KB": k["k_mlp_bf16"]["WRITE_SIZE_KB"],
                "hbm_bytes_per_launch": k["k_mlp_bf16"]["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": 12124160})
if all(x in k for x in ("trainfwd", "dgrad", "wgrad")):
    res["train_hbm_bytes_per_step"] = 2 * k["trainfwd"]["hbm_bytes_per_launch"] + 2 * k["dgrad"]["hbm_bytes_per_launch"] + k["wgrad"]["hbm_bytes_per_launch"]
    print(f"training step: {res['train_hbm_bytes_per_step'] / 1e9:.2f} GB")
json.dump(res, open(out + "/../pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/inf_* $OUT/train_* $OUT/f32_*
