#!/bin/bash
# After tools/round_runs.sh and tools/round_profiles.sh (outputs merged under gpurun_out/$R): build the committed set profiles/rNN_*.
# usage: R=r3 N=03 bash tools/collect_profiles.sh     (delete gpurun_out/$R/prof_final before the GPU run: stale files of an earlier run mix in)
set -e
R=${R:-r5}; N=${N:-05}
cd "$(dirname "$0")/.."
ks=$(ls -t gpurun_out/$R/prof_final/prof/*/*kernel_stats.csv | head -1); ds=$(ls -t gpurun_out/$R/prof_final/prof/*/*domain_stats.csv | head -1)
cp "$ks" profiles/r${N}_bench_kernel_stats.csv; cp "$ds" profiles/r${N}_bench_domain_stats.csv
python tools/profile_families.py profiles/r${N}_bench_kernel_stats.csv profiles/r${N}_families.json > /dev/null
python tools/pmc_ops.py gpurun_out/$R/prof_final/pmc r$N > /dev/null
for what in strict refiner b4; do      # (round 4: the strict mode, the refiner img2img fp8 line, the bs=4 line of config #3: same commands under rocprofv3)
  ks=$(ls -t gpurun_out/$R/prof_final/prof_$what/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$ks" ] || continue
  cp "$ks" profiles/r${N}_${what}_kernel_stats.csv
  python tools/profile_families.py profiles/r${N}_${what}_kernel_stats.csv profiles/r${N}_${what}_families.json > /dev/null
  python - "$R" "$N" "$what" <<'PY'
import json, sys
R, N, what = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/{R}/prof_final/prof_{what}_bench.json").read().strip().splitlines()[-1])
json.dump(d, open(f"profiles/r{N}_{what}_bench_line.json", "w"), indent=1)
PY
done
python - "$R" "$N" <<'PY'
import glob, json, os, sys
R, N = sys.argv[1], sys.argv[2]
last = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
json.dump(last(f"gpurun_out/{R}/prof_final/prof_bench.json"), open(f"profiles/r{N}_bench_line.json", "w"), indent=1)
json.dump({d: json.load(open(f"gpurun_out/{R}/prof_final/vendor_{d}.json")) for d in ("bf16", "fp16")}, open(f"profiles/r{N}_vendor.json", "w"), indent=1)
out = {}
for fn in sorted(glob.glob(f"gpurun_out/{R}/final/bench_*.json")):
    try:
        out[os.path.basename(fn)[6:-5]] = last(fn)
    except Exception as e:
        print("skipped", fn, e)
json.dump(out, open(f"profiles/r{N}_bench_lines.json", "w"), indent=1)
for k, d in out.items():
    print(k, d["value"], d["ms_per_step"], d["dtype"], d["n_gpus"])
PY
