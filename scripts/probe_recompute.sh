#!/bin/bash
# round 5: the measured go / no-go of "store every other trunk layer, recompute the skipped one inside the weight-gradient
# kernel".  Timing builds (WRONG results by construction; build.py demands the opt-in):
#   MIPNERF_EXPERIMENT_BUILD=1 MLP_TRAIN_SKIP_STORES=0,2,4,6                                   MIPNERF_LIB_NAME=libmipnerf_hip_sk.so    python -m mipnerf_pl_amd.build
#   MIPNERF_EXPERIMENT_BUILD=1 MLP_TRAIN_SKIP_STORES=0,2,4,6 MLP_WGRAD_RECOMPUTE_PROBE=298     MIPNERF_LIB_NAME=libmipnerf_hip_skrc.so  python -m mipnerf_pl_amd.build
#   ... MLP_WGRAD_RECOMPUTE_SCHED=1 (two accumulator chains, operands four k-steps ahead)      MIPNERF_LIB_NAME=libmipnerf_hip_skrc2.so python -m mipnerf_pl_amd.build
#   python -m mipnerf_pl_amd.build          # LAST: restores the tracked generated sources and the product library
# sk = the training forward skips the T-block stores of x1, x3, x5, x7 and nothing pays for it (the upper bound of the gain); skrc = the
# weight-gradient jobs L1, L3, L5, L7 (bit mask 298) additionally carry the recompute's instruction mix.  Alternating timing of the captured
# training step, per-kernel times (rocprofv3 --kernel-trace --stats), HBM bytes and cycles (separate --pmc passes).
# -> profiles/r05b_recompute_probe_*.txt, r05b_train_*_kernel_stats.csv, r05c_recompute_probe_*.txt
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r05}
export TMPDIR=/tmp MIPNERF_ALLOW_EXPERIMENT_LIB=1 MIPNERF_ZERO_SCRATCH=1
C=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc
LIBS="libmipnerf_hip.so libmipnerf_hip_sk.so libmipnerf_hip_skrc.so libmipnerf_hip_skrc2.so"
for i in 1 2 3; do
  for lib in $LIBS; do
    [ -f $C/$lib ] || continue
    MIPNERF_LIB=$C/$lib timeout 200 python bench.py --mode train --steps 50 --warmup 5 --no-cpu-baseline --preheat-seconds 2 2>gpurun_out/ab_train.err | python -c "
import sys, json
ls = [x for x in sys.stdin if x.startswith('{')]
if not ls: print('$lib', 'no line (non-finite loss?)')
else:
    l = json.loads(ls[-1]); print('$lib', l['ms_per_step'])"
  done
done | tee gpurun_out/${T}_recompute_probe_ab.txt
cd /tmp
for lib in $LIBS; do
  [ -f $C/$lib ] || continue
  MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t -o train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 30 --warmup 5 --no-graph --no-cpu-baseline --preheat-seconds 2 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_train_${lib%.so}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t
done
for lib in libmipnerf_hip.so libmipnerf_hip_skrc.so libmipnerf_hip_skrc2.so; do
  [ -f $C/$lib ] || continue
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $c | tr ' ' '_')
    MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_t/${lib%.so}__$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-graph --no-cpu-baseline --preheat-seconds 0 > /dev/null 2>&1; echo "pmc $lib $tag rc=$?"
  done
done
python - $GRAFT_REPO_ROOT/gpurun_out/pmc_t <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${T}_recompute_probe_counters.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
trn = lambda n: "wgrad" if "k_mlp_wgrad" in n else "trainfwd" if "trainfwd" in n else "dgrad" if "dgrad" in n else None
libs = sorted({os.path.basename(d).split("__")[0] for d in glob.glob(out + "/*__*")})
for lib in libs:
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{lib}__*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = trn(r["Kernel_Name"])
            if k:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    step = 0.0
    for k, d in sorted(vals.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        b = m.get("FETCH_SIZE", 0) * 1024 * 2.0 + m.get("WRITE_SIZE", 0) * 1024         # FETCH x 2 on gfx950 (MI355X_MICROARCH.md)
        step += b * (1 if k == "wgrad" else 2)
        print(f"{lib} {k}: {b / 1e9:.3f} GB per launch; cycles " + " ".join(f"{c}={x:.4g}" for c, x in sorted(m.items()) if "SIZE" not in c))
    print(f"{lib} training step: {step / 1e9:.2f} GB")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_t
