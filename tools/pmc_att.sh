#!/bin/bash
root=$PWD
out=$root/gpurun_out/r2/pmc_att
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  d=$out/$tag; mkdir -p $d
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $root/tools/one_op.py attn 1 4096 4096 10 > $d/log.txt 2>&1 || echo FAIL $tag
  find $d -name "*kernel_trace.csv" -delete
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$out/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'attn' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(f"{k:36s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
