#!/bin/bash
# round 5, second GPU call: (1) parity suite on the final form of the ray-kernel changes, (2) kernel stats of the headline forward for the DPP
# and the shuffle build, (3) the measured go / no-go of "store every other layer + recompute in the weight-gradient kernel" (VERDICT r04 #2):
# timing builds libmipnerf_hip_sk.so (the training forward skips the T-block stores of x1, x3, x5, x7) and libmipnerf_hip_skrc.so (+ the
# weight-gradient jobs of those layers carry the recompute's instruction mix), alternating against the product library, with per-kernel
# times (rocprofv3 --kernel-trace --stats) and HBM bytes (separate --pmc passes)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r05b}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/${T}_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -k "not trained_unbounded_field" 2>&1 | tail -12 > gpurun_out/${T}_pytest_gpu_tail.txt
tail -6 gpurun_out/${T}_pytest_gpu_tail.txt
cp gpurun_out/parity.jsonl gpurun_out/${T}_parity.jsonl
C=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc
cd /tmp
for lib in libmipnerf_hip.so libmipnerf_hip_shfl.so; do
  MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 20 --ceiling-seconds 0 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_inference_${lib%.so}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b
  echo "== $lib"; head -5 $GRAFT_REPO_ROOT/gpurun_out/${T}_inference_${lib%.so}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-50,120-300
done
cd $GRAFT_REPO_ROOT
export MIPNERF_ALLOW_EXPERIMENT_LIB=1 MIPNERF_ZERO_SCRATCH=1
LIBS="libmipnerf_hip.so libmipnerf_hip_sk.so libmipnerf_hip_skrc.so"
for i in 1 2 3; do
  for lib in $LIBS; do
    MIPNERF_LIB=$C/$lib timeout 200 python bench.py --mode train --steps 50 --warmup 5 --no-cpu-baseline --preheat-seconds 2 2>gpurun_out/ab_train.err | python -c "
import sys, json
ls = [x for x in sys.stdin if x.startswith('{')]
if not ls: print('$lib', 'no line (non-finite loss?)')
else:
    l = json.loads(ls[-1]); print('$lib', l['ms_per_step'])"
  done
done | tee gpurun_out/${T}_recompute_probe_ab.txt
tail -3 gpurun_out/ab_train.err
cd /tmp
for lib in $LIBS; do
  MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t -o train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 30 --warmup 5 --no-graph --no-cpu-baseline --preheat-seconds 2 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_train_${lib%.so}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t
  echo "== $lib"; head -6 $GRAFT_REPO_ROOT/gpurun_out/${T}_train_${lib%.so}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-300
done
for lib in libmipnerf_hip.so libmipnerf_hip_skrc.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_t/${lib%.so}_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-graph --no-cpu-baseline --preheat-seconds 0 > /dev/null 2>&1; echo "pmc $lib $c rc=$?"
  done
done
python - $GRAFT_REPO_ROOT/gpurun_out/pmc_t <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${T}_recompute_probe_traffic.txt
import csv, glob, sys, collections
out = sys.argv[1]
trn = lambda n: "wgrad" if "k_mlp_wgrad" in n else "trainfwd" if "trainfwd" in n else "dgrad" if "dgrad" in n else None
for lib in ("libmipnerf_hip", "libmipnerf_hip_skrc"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/{lib}_{c}/**/*counter_collection.csv", recursive=True)
        acc = collections.defaultdict(list)
        if f:
            for r in csv.DictReader(open(f[0])):
                k = trn(r["Kernel_Name"])
                if k:
                    acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            tot.setdefault(k, {})[c] = sum(v) / len(v)
    step = 0.0
    for k, d in sorted(tot.items()):
        b = d.get("FETCH_SIZE", 0) * 1024 * 2.0 + d.get("WRITE_SIZE", 0) * 1024        # FETCH x 2 on gfx950 (MI355X_MICROARCH.md)
        step += b * (1 if k == "wgrad" else 2)
        print(f"{lib} {k}: FETCH {d.get('FETCH_SIZE', 0):.6g} KB x2 + WRITE {d.get('WRITE_SIZE', 0):.6g} KB = {b / 1e9:.3f} GB per launch")
    print(f"{lib} training step: {step / 1e9:.2f} GB")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_t
