python scripts/micro/dbg_adam.py 2>&1 | grep -v amdgpu.ids | head -8
python bench.py --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 10 > $OUT/bench_fp32.json 2>/dev/null; tail -c 700 $OUT/bench_fp32.json
python scripts/micro/cpu_threads.py 2>&1 | grep -v amdgpu.ids
