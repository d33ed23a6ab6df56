set -x
cd $GRAFT_REPO_ROOT; o=gpurun_out/r5; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $o/t_full.log 2>&1; echo rc=$?
tail -5 $o/t_full.log
python tools/launch_floor.py > $o/launch_floor.log 2>&1; cat $o/launch_floor.log | grep -v amdgpu
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -5
