# A/B: 8-wave workgroups (1 per CU) vs 4-wave workgroups (2 per CU) of the inference MLP kernel
for i in 1 2; do
  python scripts/prof_mlp.py --iters 40
  MIPNERF_LIB=$PWD/mipnerf_pl_amd/csrc/libmipnerf_hip_w4.so python scripts/prof_mlp.py --iters 40
done
MIPNERF_LIB=$PWD/mipnerf_pl_amd/csrc/libmipnerf_hip_w4.so python bench.py --mode inference --no-cpu-baseline > $OUT/bench_w4.json 2>/dev/null; tail -c 1500 $OUT/bench_w4.json
MIPNERF_LIB=$PWD/mipnerf_pl_amd/csrc/libmipnerf_hip_w4.so python -m pytest tests/test_gpu_forward.py -q -x 2>&1 | tail -3
# kernel trace of the default bench
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sustain-seconds 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; rm -rf $OUT/prof
head -12 $OUT/kernel_stats.csv | cut -c1-160
