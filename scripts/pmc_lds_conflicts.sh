#!/bin/bash
# round 5: LDS bank-conflict cycles of every MLP kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, own --pmc pass): the counter that found the
# 4-way conflicts of the fragment-sourced weight-gradient jobs.  Targets: the headline forward, the standard training step, the unbounded
# model's forward and one-graph training step.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lds; mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/$name -o pmc -- "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fwd python $ROOT/scripts/prof_fwd.py
run train python $ROOT/bench.py --mode train --no-graph --steps 5 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0.2
run fwd360 python $ROOT/scripts/micro/prof_unbounded.py bf16 2
run train360 python $ROOT/scripts/micro/prof_train360.py bf16_graph
python - $OUT <<'PY' | tee $ROOT/gpurun_out/${TAG:-r05}_lds_conflicts.txt
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("fwd", "train", "fwd360", "train360"):
    fs = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "mip" not in k:
            continue
        short = k.split("(")[0][-46:]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if m.get("SQ_LDS_IDX_ACTIVE", 0) < 1e5:
            continue
        print(f"{name:9s} {k:46s} LDS active {m['SQ_LDS_IDX_ACTIVE']:.3e}  bank-conflict {m.get('SQ_LDS_BANK_CONFLICT', 0):.3e}  "
              f"= {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f} of the active LDS cycles")
PY
rm -rf $OUT
