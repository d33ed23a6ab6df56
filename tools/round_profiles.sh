# The round's profile set (run on the GPU box from the repo root): rocprofv3 kernel-trace summary of the bench command, the
# per-shape PMC passes, the vendor comparison.  Outputs under gpurun_out/; tools/profile_families.py and tools/pmc_ops.py
# turn them into profiles/.
R=${R:-r5}
set -x
out=gpurun_out/$R/prof_final
rm -rf $out && mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 50 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err
cd $GRAFT_REPO_ROOT/$out/prof && find . -name "*kernel_trace.csv" -delete
# the same for the strict mode (fp32 storage, split operands) and for the refiner img2img line of config #5
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_strict -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 50 --warmup 10 --mode step --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/prof_strict_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_strict_bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_refiner -- python3 $GRAFT_REPO_ROOT/bench.py --model refiner --img2img 0.3 --steps 30 --warmup 15 --no-cpu-baseline --no-extras --fp8 > $GRAFT_REPO_ROOT/$out/prof_refiner_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_refiner_bench.err
# (config #3: the bs=4 line)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_b4 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-extras  > $GRAFT_REPO_ROOT/$out/prof_b4_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_b4_bench.err
find $GRAFT_REPO_ROOT/$out/prof_strict $GRAFT_REPO_ROOT/$out/prof_refiner $GRAFT_REPO_ROOT/$out/prof_b4 -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
python tools/gemm_vs_vendor.py $out/vendor_bf16.json bf16 > $out/vendor_bf16.log 2>&1
python tools/gemm_vs_vendor.py $out/vendor_fp16.json fp16 > $out/vendor_fp16.log 2>&1
bash tools/pmc_ops.sh $out/pmc > $out/pmc.log 2>&1
tail -2 $out/pmc.log
