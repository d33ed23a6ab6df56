# strict (fp32 / split) mode: the split tests, the fp32 cases of the per-op suite, then the strict bench line with the shape census
set -x
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4/strict1; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_split_gpu.py -q > $o/split_tests.log 2>&1; echo "split rc=$?" >> $o/split_tests.log; tail -15 $o/split_tests.log
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -q -k "dtype0 or fp32 or float32" > $o/ops_fp32.log 2>&1; echo "ops rc=$?" >> $o/ops_fp32.log; tail -15 $o/ops_fp32.log
ST_CENSUS_SHAPES=1 timeout -k 10 600 python bench.py --dtype fp32 --steps 5 --warmup 2 --mode step --no-cpu-baseline --no-extras > $o/bench.json 2> $o/census.err; tail -3 $o/census.err; cat $o/bench.json | cut -c1-300
