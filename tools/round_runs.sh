set -x
mkdir -p gpurun_out/r2/final && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r2/final/bench_default.json 2> gpurun_out/r2/final/bench_default.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 2 > gpurun_out/r2/final/bench_b2.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 > gpurun_out/r2/final/bench_b4.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --fp8 > gpurun_out/r2/final/bench_b4_fp8.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --fp8 > gpurun_out/r2/final/bench_b1_fp8.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof_bench.err
cd $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof && find . -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
bash tools/pmc_ops.sh gpurun_out/r2/final/pmc > gpurun_out/r2/final/pmc.log 2>&1
tail -2 gpurun_out/r2/final/pmc.log
for f in gpurun_out/r2/final/bench_*.json gpurun_out/r2/final/prof_bench.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("cpu_baseline"), d["roofline"]["frac"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
