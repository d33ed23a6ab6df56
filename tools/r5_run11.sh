set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 700 python -m pytest tests/test_fp8_gpu.py tests/test_hooks_gpu.py -x -q -k "fp8 or tiny" > $o/t11.log 2>&1 || { tail -30 $o/t11.log; exit 1; }
tail -2 $o/t11.log
ST_VARIANT=dev ST_NO_COLSPLIT=1 timeout -k 10 200 python tools/colsplit_ab.py 2>&1 | grep "^M=" > $o/colsplit_off.log; cat $o/colsplit_off.log
ST_VARIANT=dev timeout -k 10 200 python tools/colsplit_ab.py 2>&1 | grep "^M=" > $o/colsplit_on.log; cat $o/colsplit_on.log
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 2 > $o/bench_b2d.json 2> $o/bench_b2d.err || { tail -5 $o/bench_b2d.err; exit 1; }
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 4 > $o/bench_b4d.json 2> $o/bench_b4d.err
timeout -k 10 200 python bench.py --steps 50 --warmup 50 --no-cpu-baseline --no-extras > $o/bench_b1d.json 2> $o/bench_b1d.err
python - <<'PY'
import json
for f in ("bench_b1d","bench_b2d","bench_b4d"):
    try:
        d=json.loads(open(f"gpurun_out/r5/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
