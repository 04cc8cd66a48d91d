#!/bin/bash
# round 4: rocprofv3 kernel stats of the fp32 configs[3] forward (register-resident kernel) and of the default bench modes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 0 > $OUT/rocprof_f32.log 2>&1
f=$(find $OUT/prof_f32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04m_fp32_c4_kernel_stats.csv
rm -rf $OUT/prof_f32
head -8 $OUT/r04m_fp32_c4_kernel_stats.csv | cut -c1-200
