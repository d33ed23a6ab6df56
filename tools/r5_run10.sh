set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -x -q -k "linear or hints or fp8" > $o/t10.log 2>&1 || { tail -30 $o/t10.log; exit 1; }
tail -2 $o/t10.log
timeout -k 10 300 python tools/gemm_vs_vendor.py $o/vendor_c.json > $o/vendor_c.log 2>&1; grep "B=" $o/vendor_c.log
timeout -k 10 200 python bench.py --steps 50 --warmup 50 --no-cpu-baseline --no-extras > $o/bench_b1c.json 2> $o/bench_b1c.err || { tail -5 $o/bench_b1c.err; exit 1; }
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 2 > $o/bench_b2c.json 2> $o/bench_b2c.err
python - <<'PY'
import json
for f in ("bench_b1c","bench_b2c"):
    try:
        d=json.loads(open(f"gpurun_out/r5/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
