#!/bin/bash
# Developer helper: build an alternative library under tools/_variants/<name>/ with extra -D flags.
# usage: tools/build_variant.sh <name> [-DFLAG ...]    (tools/devlib.use_variant(<name>) / ST_VARIANT=<name> in the tools selects it)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/_variants/$name
mkdir -p "$out/obj"
objs=()
for f in "$root"/stabletriton_amd/csrc/*.hip "$root"/tools/dev_kernels/*.hip; do
  o=$out/obj/$(basename "${f%.hip}").o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "$@" -c "$f" -o "$o" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc "${objs[@]}" -o "$out/libstabletriton_amd.so"
echo "$out/libstabletriton_amd.so"
