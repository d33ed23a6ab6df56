set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_hooks_gpu.py -x -q -s -k "timestep or f3_euler50_fp32 or callsite_sdxl_fp32 or f1_ or callsite_tiny or tiny_unet_step or modes or f3_b2" > $o/t4.log 2>&1; echo rc=$?
grep -v "^$" $o/t4.log | grep -i "F3\|F1\|passed\|failed\|error\|assert\|tiny" | head -60
