#!/bin/bash
# usage: tools/pmc_one.sh "<counters>" <outtag> -- python3 tools/one_op.py ...   (developer: one PMC pass, prints per-kernel averages)
ctr=$1; tag=$2; shift 3
root=$PWD
out=$root/gpurun_out/pmc_one/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- "$@" > $out/log.txt 2>&1 || echo FAIL
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if any(x in k for x in ('attn','gemm','conv','gn_')):
        print(k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
