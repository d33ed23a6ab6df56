set -x
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4/strict0; mkdir -p $o
ST_CENSUS_SHAPES=1 python bench.py --dtype fp32 --steps 5 --warmup 2 --mode step --no-cpu-baseline --no-extras > $o/bench.json 2> $o/census.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 10 --warmup 2 --mode step --no-cpu-baseline --no-extras --no-census > $GRAFT_REPO_ROOT/$o/prof_bench.json 2> $GRAFT_REPO_ROOT/$o/prof_bench.err
cd $GRAFT_REPO_ROOT/$o/prof && find . -name "*kernel_trace.csv" -delete
ls -R $GRAFT_REPO_ROOT/$o | head
