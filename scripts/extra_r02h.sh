python bench.py --precision fp32 --mode train --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_fp32_train.json 2>$OUT/bench_fp32_train.err; python -c "
import json; d=json.load(open('$OUT/bench_fp32_train.json')); print('fp32 train ms', d['ms_per_step'])"; tail -2 $OUT/bench_fp32_train.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --mode train --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/fp32_train_kernel_stats.csv; rm -rf $OUT/prof
head -8 $OUT/fp32_train_kernel_stats.csv | cut -c1-150
