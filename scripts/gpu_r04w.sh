#!/bin/bash
# round 4: first GPU run of the bf16 kernels of the unbounded-scene model (k_pre_gemm + trunk): parity tests, timing, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_unbounded_bf16.py -q -m gpu -x 2>&1 | tail -25 > $OUT/r04w_pytest.txt
cat $OUT/r04w_pytest.txt
timeout 120 python scripts/micro/prof_unbounded.py bf16 10 2>&1 | tail -3 | tee $OUT/r04w_timing.txt
timeout 120 python scripts/micro/prof_unbounded.py fp32 4 2>&1 | tail -1 | tee -a $OUT/r04w_timing.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_u16 -o u16 -- python $GRAFT_REPO_ROOT/scripts/micro/prof_unbounded.py bf16 6 > $OUT/rocprof_u16.log 2>&1
f=$(find $OUT/prof_u16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04w_unbounded_bf16_kernel_stats.csv
rm -rf $OUT/prof_u16
head -8 $OUT/r04w_unbounded_bf16_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_unbounded.py tests/test_gpu_stages.py tests/test_gpu_f32r.py -q -m gpu -x 2>&1 | tail -4 | tee $OUT/r04w_pytest_regress.txt
