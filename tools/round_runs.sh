# The round's bench lines (run on the GPU box from the repo root; outputs under gpurun_out/$R/final): the default call with
# its cpu_baseline and side measurements, the driver's call, the other batch sizes, fp16, the fp8 mode, the two-rank gloo
# rehearsal of the multi-GPU path on one card; since round 4 the strict (fp32 / split operands) mode and the refiner img2img lines of config #5.  tools/round_profiles.sh collects the rocprofv3 / PMC set.
R=${R:-r5}
set -x
mkdir -p gpurun_out/$R/final && cd $GRAFT_REPO_ROOT
o=gpurun_out/$R/final
python bench.py > $o/bench_default.json 2> $o/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_driver_call.json 2>/dev/null
python bench.py --steps 50 --warmup 50 --no-cpu-baseline --no-extras --dtype fp16 > $o/bench_fp16.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 2 > $o/bench_b2.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 2 --fp8 > $o/bench_b2_fp8.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 4 > $o/bench_b4.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 4 --dtype fp16 > $o/bench_b4_fp16.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch 4 --fp8 > $o/bench_b4_fp8.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --fp8 > $o/bench_b1_fp8.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --mode step --no-cpu-baseline --no-extras --dtype fp32 > $o/bench_strict_fp32.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --mode step --no-cpu-baseline --no-extras --dtype fp32 --batch 4 > $o/bench_strict_fp32_b4.json 2>/dev/null
python bench.py --model refiner --img2img 0.3 --steps 30 --warmup 15 --no-cpu-baseline --no-extras > $o/bench_refiner_img2img.json 2>/dev/null
python bench.py --model refiner --img2img 0.3 --steps 30 --warmup 15 --no-cpu-baseline --no-extras --fp8 > $o/bench_refiner_img2img_fp8.json 2>/dev/null
python bench.py --gpus 2 --same-device --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $o/bench_2rank_gloo_same_device.json 2> $o/bench_2rank.err
# the driver's own launcher for N > 1 (torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from its env; fresh children, nothing here has touched the GPU)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --same-device > $o/bench_2rank_torchrun_same_device.json 2> $o/bench_2rank_torchrun.err
for f in $o/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["dtype"], d.get("roofline",{}).get("frac"), {k:d.get(k) for k in ("loop_graph_it_per_s","other16_it_per_s","strict_fp32_it_per_s")})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
