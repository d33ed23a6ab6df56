set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -x -q -k "linear or abi or hints" > $o/t2.log 2>&1 || { tail -30 $o/t2.log; exit 1; }
tail -3 $o/t2.log
ST_VARIANT=probe8 timeout -k 10 200 python tools/gemm8p_probe.py > $o/probe_direct.log 2>&1
ST_VARIANT=probe8 ST_8P_STAGED=1 timeout -k 10 200 python tools/gemm8p_probe.py > $o/probe_staged.log 2>&1
cat $o/probe_direct.log $o/probe_staged.log
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $o/bench_b1.json 2> $o/bench_b1.err || { tail -5 $o/bench_b1.err; exit 1; }
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --batch 4 > $o/bench_b4.json 2> $o/bench_b4.err || { tail -5 $o/bench_b4.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_b1","bench_b4"):
    try:
        d=json.loads(open(f"gpurun_out/r5/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/vendor_trace -o vt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/vendor_names.py > $GRAFT_REPO_ROOT/$o/vendor_trace.log 2>&1)
find $o/vendor_trace -name "*kernel_stats*" | head; f=$(find $o/vendor_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-400 "$f" | head -30
