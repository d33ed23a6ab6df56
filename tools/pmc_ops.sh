#!/bin/bash
# Hardware counters per operator shape (bench.py's full-model run crashes under --pmc in this ROCm build, so the shapes
# are profiled one at a time).  Run on the GPU box from the repo root:  bash tools/pmc_ops.sh [outdir]
# Three separate passes per shape (the TCC counters do not fit one pass; never combined with --stats / other traces):
#   SQ: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE   TCC: FETCH_SIZE   TCC: WRITE_SIZE
# tools/pmc_ops.py turns the CSVs into profiles/rNN_traffic.json.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$root/gpurun_out/pmc_ops}
case "$out" in /*) ;; *) out=$root/$out ;; esac
mkdir -p "$out"
hipcc -O3 -w --offload-arch=gfx950 "$root/tools/micro/mfma_peak.hip" -o /tmp/mfma_peak
cd /tmp && export TMPDIR=/tmp
pass() {   # name counters... -- program args
  local name=$1; shift
  local ctr=$1; shift
  local tag=$1; shift
  local d=$out/${name}__$tag
  mkdir -p "$d"
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d" -- "$@" > "$d/log.txt" 2>&1 || echo "FAILED $name $tag" >> "$out/failures.txt"
  find "$d" -name "*kernel_trace.csv" -delete
  find "$d" -name "*agent_info.csv" -delete
}
pass calib "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" sq /tmp/mfma_peak 20000
while read -r name args; do
  [ -z "$name" ] && continue
  pass "$name" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" sq python3 "$root/tools/one_op.py" $args
  pass "$name" "FETCH_SIZE" fetch python3 "$root/tools/one_op.py" $args
  pass "$name" "WRITE_SIZE" write python3 "$root/tools/one_op.py" $args
  echo "$name done"
done <<'SHAPES'
attn_self_4096 attn 1 4096 4096 10
attn_self_1024 attn 1 1024 1024 20
attn_cross_4096 attn 1 4096 77 10
attn_cross_1024 attn 1 1024 77 20
attn_self_1024_b4 attn 4 1024 1024 20
xattn_1024x1280 xattn 1 1024 1280 20 77
xattn_4096x640 xattn 1 4096 640 10 77
attn_self_4096_b4 attn 4 4096 4096 10
linear_1024x1280x5120_lng linear 1024 1280 5120 lng
linear_1024x5120x1280 linear 1024 5120 1280
linear_1024x1280x1280 linear 1024 1280 1280
linear_1024x1280x3840_ln linear 1024 1280 3840 ln
linear_4096x640x640 linear 4096 640 640
linear_4096x640x2560_lng linear 4096 640 2560 lng
linear_4096x2560x640 linear 4096 2560 640
linear_4096x1280x5120_lng_b4 linear 4096 1280 5120 lng
conv_1280_32 conv 1 1280 32 1280 3 1 0
conv_640_64 conv 1 640 64 640 3 1 0
conv_320_128 conv 1 320 128 320 3 1 0
gn_320_128 gn 1 320 128 1
gn_1280_32 gn 1 1280 32 1
gn_640_64 gn 1 640 64 1
SHAPES
echo all done
