#!/bin/bash
# round 3: the four in-process MFMA ceilings and the training kernels' times ON THE SAME BOX (is the training forward at the
# ceiling of an MFMA stream that also writes its T-blocks?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/micro/ceiling_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z2_ceiling_variants.txt
cd /tmp
rm -rf /tmp/prof_tr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --mode train --no-graph --steps 30 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r03z2_train_prof.json" 2>/dev/null
f=$(find /tmp/prof_tr -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/r03z2_train_kernel_stats.csv"
head -6 "$f" | cut -c1-160
cd "$GRAFT_REPO_ROOT"
python scripts/micro/ceiling_variants.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03z2_ceiling_variants.txt
