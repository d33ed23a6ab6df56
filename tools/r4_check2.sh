# round-4 check 2: ADVICE fixes (fp8 determinism, hooks cache), bench default line with the strict entry, the refiner img2img lines
set -x
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4/check2; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_hooks_gpu.py -q -x > $o/tests.log 2>&1; echo "rc=$?" >> $o/tests.log; tail -5 $o/tests.log
timeout -k 10 900 python bench.py --no-cpu-baseline > $o/bench_default.json 2> $o/bench_default.err; tail -2 $o/bench_default.err; cut -c1-200 $o/bench_default.json
timeout -k 10 600 python bench.py --model refiner --img2img 0.3 --steps 30 --warmup 15 --no-cpu-baseline --no-extras > $o/bench_refiner.json 2> $o/bench_refiner.err; tail -2 $o/bench_refiner.err; cut -c1-300 $o/bench_refiner.json
timeout -k 10 600 python bench.py --model refiner --img2img 0.3 --steps 30 --warmup 15 --no-cpu-baseline --no-extras --fp8 > $o/bench_refiner_fp8.json 2> $o/bench_refiner_fp8.err; tail -2 $o/bench_refiner_fp8.err; cut -c1-300 $o/bench_refiner_fp8.json
