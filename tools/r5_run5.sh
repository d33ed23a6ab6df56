set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
ST_VARIANT=dev timeout -k 10 300 python tools/gemm4w_time.py 100 102 > $o/g4w.log 2>&1; cat $o/g4w.log | grep -v amdgpu
ST_VARIANT=dev timeout -k 10 400 python tools/gemm_sweep.py 4 all > $o/sweep_b4.log 2>&1; grep "^M=" $o/sweep_b4.log
ST_VARIANT=dev ST_BENCH_DTYPE=fp32 timeout -k 10 400 python tools/gemm_sweep.py 1 all > $o/sweep_strict_b1.log 2>&1; grep "^M=" $o/sweep_strict_b1.log
