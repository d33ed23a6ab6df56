# round 5, first GPU call: the register epilogue + rebalanced phases of the eight-phase kernel (tests, vendor table, same-box A/B
# against the staged epilogue through the developer build's ST_8P_STAGED knob), the bs=1 sweep with the 64 x 80 tile, bench lines
set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -x -q -k "linear or abi" > $o/t1.log 2>&1 || { tail -30 $o/t1.log; exit 1; }
tail -3 $o/t1.log
timeout -k 10 300 python tools/gemm_vs_vendor.py $o/vendor_a.json > $o/vendor_a.log 2>&1 && grep -c "vendor/ours" $o/vendor_a.log
ST_VARIANT=dev ST_8P_STAGED=1 timeout -k 10 300 python tools/gemm_vs_vendor.py $o/vendor_staged.json > $o/vendor_staged.log 2>&1
ST_VARIANT=dev timeout -k 10 300 python tools/gemm_sweep.py 1 plain > $o/sweep_b1.log 2>&1
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $o/bench_b1.json 2> $o/bench_b1.err
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --batch 4 > $o/bench_b4.json 2> $o/bench_b4.err
python - <<'PY'
import json
for f in ("bench_b1","bench_b4"):
    try:
        d=json.loads(open(f"gpurun_out/r5/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
