#!/bin/bash
# Developer helper: rebuild ONE source with extra -D flags and link it with the product objects of the other sources.
# usage: tools/build_one_variant.sh <name> <source.hip> [-DFLAG ...]   -> tools/_variants/<name>/libstabletriton_amd.so
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/_variants/$name
mkdir -p "$out"
base=$(basename "${src%.hip}")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "$@" -c "$root/stabletriton_amd/csrc/$base.hip" -o "$out/$base.o"
objs=("$out/$base.o")
for f in "$root"/stabletriton_amd/csrc/*.hip; do
  b=$(basename "${f%.hip}")
  [ "$b" = "$base" ] || objs+=("$root/stabletriton_amd/lib/obj/$b.o")
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc "${objs[@]}" -o "$out/libstabletriton_amd.so"
rm -f "$out/$base.o"
echo "$out/libstabletriton_amd.so"
