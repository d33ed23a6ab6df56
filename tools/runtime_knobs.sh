#!/bin/bash
# Developer experiment: the bench line (loop-mode hipGraph replay, headline config) under HIP runtime environment knobs that
# touch kernel dispatch / graph replay.  One bench process per setting, sequentially.  usage: tools/runtime_knobs.sh [outfile]
out=${1:-gpurun_out/r4/runtime_knobs.txt}
mkdir -p "$(dirname "$out")"
run() {
  local tag=$1; shift
  local line
  line=$(env "$@" timeout -k 10 240 python bench.py --no-extras --no-cpu-baseline --no-census --steps 100 --warmup 50 2>/dev/null | tail -1)
  python3 - "$tag" "$line" >> "$out" <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(f"{tag:60s} {d['value']:8.3f} it/s  {d['ms_per_step']:7.3f} ms")
except Exception as e:
    print(f"{tag:60s} failed: {line[:120]!r}")
PY
}
: > "$out"
run "default" ST_NOP=1
run "default (again)" ST_NOP=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run "DEBUG_HIP_GRAPH_BATCH_SIZE=16" DEBUG_HIP_GRAPH_BATCH_SIZE=16
run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run "ROC_USE_FGS_KERNARG=0" ROC_USE_FGS_KERNARG=0
run "AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
run "ROC_SYSTEM_SCOPE_SIGNAL=0" ROC_SYSTEM_SCOPE_SIGNAL=0
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "default (last)" ST_NOP=1
cat "$out"
