# The round's bench lines (run on the GPU box from the repo root; outputs under gpurun_out/r2/final): the default call with
# its cpu_baseline, the other batch sizes and the fp8 mode.  tools/round_profiles.sh collects the rocprofv3 / PMC set.
set -x
mkdir -p gpurun_out/r2/final && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r2/final/bench_default.json 2> gpurun_out/r2/final/bench_default.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 2 > gpurun_out/r2/final/bench_b2.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 > gpurun_out/r2/final/bench_b4.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --fp8 > gpurun_out/r2/final/bench_b4_fp8.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --fp8 > gpurun_out/r2/final/bench_b1_fp8.json 2>/dev/null
for f in gpurun_out/r2/final/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("cpu_baseline"), d["roofline"]["frac"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
