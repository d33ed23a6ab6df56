#!/bin/bash
# Developer helper: rebuild SOME sources with extra -D flags and link them with the variant's own earlier objects (or,
# where it has none, the product objects of the other sources).
# usage: tools/build_one_variant.sh <name> <source.hip>... [-DFLAG ...]   -> tools/_variants/<name>/libstabletriton_amd.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/_variants/$name
mkdir -p "$out/obj"
srcs=(); flags=(); pids=()
for a in "$@"; do case "$a" in *.hip) srcs+=("$a");; *) flags+=("$a");; esac; done
for src in "${srcs[@]}"; do
  base=$(basename "${src%.hip}")
  srcfile="$root/stabletriton_amd/csrc/$base.hip"; [ -f "$srcfile" ] || srcfile="$root/tools/dev_kernels/$base.hip"      # (developer kernels: tools/dev_kernels/)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "${flags[@]}" -c "$srcfile" -o "$out/obj/$base.o.new" &
  pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid" || { echo "compile failed"; exit 1; }; done      # (a failed compile must not link a stale object)
for src in "${srcs[@]}"; do base=$(basename "${src%.hip}"); mv "$out/obj/$base.o.new" "$out/obj/$base.o"; done
objs=()
for f in "$root"/stabletriton_amd/csrc/*.hip; do
  b=$(basename "${f%.hip}")
  if [ -f "$out/obj/$b.o" ]; then objs+=("$out/obj/$b.o"); else objs+=("$root/stabletriton_amd/lib/obj/$b.o"); fi
done
for f in "$root"/tools/dev_kernels/*.hip; do b=$(basename "${f%.hip}"); [ -f "$out/obj/$b.o" ] && objs+=("$out/obj/$b.o"); done      # developer kernels this variant built
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc "${objs[@]}" -o "$out/libstabletriton_amd.so"
echo "$out/libstabletriton_amd.so"
