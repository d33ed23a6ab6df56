set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
ST_VARIANT=probe8 timeout -k 10 200 python tools/gemm8p_probe.py > $o/probe2_direct.log 2>&1
ST_VARIANT=probe8 ST_8P_STAGED=1 timeout -k 10 200 python tools/gemm8p_probe.py > $o/probe2_staged.log 2>&1
cat $o/probe2_direct.log $o/probe2_staged.log
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py tests/test_hooks_gpu.py tests/test_torch_ops.py tests/test_ops_gpu.py -x -q -s -k "f2_large or test_group_norm or f3_euler50_fp32 or callsite_sdxl_fp32 or dispatcher or f3_euler50_bf16 or f1_ or callsite_sdxl_fp16" > $o/t3.log 2>&1; echo rc=$?
grep -v "^$" $o/t3.log | grep -i "F3\|F2-large\|F1\|passed\|failed\|error\|assert" | head -60
