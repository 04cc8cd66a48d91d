python scripts/micro/dbg_adam.py 2>&1 | grep -v amdgpu.ids
python bench.py --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 10 > $OUT/bench_fp32.json 2>/dev/null; tail -c 1200 $OUT/bench_fp32.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sustain-seconds 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
head -14 $OUT/kernel_stats.csv | cut -c1-180
