"""Developer check: registers, LDS and scratch of every kernel in the built library (from the code-object metadata).
usage: python tools/kernel_resources.py [filter [library.so]]   - prints kernels with scratch first."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "stabletriton_amd", "lib", "libstabletriton_amd.so")
tmp = tempfile.mkdtemp()
# the .so embeds one fat binary per object; llvm-objdump --offloading extracts them NEXT TO its input: work on a copy
import shutil
shutil.copy(lib, os.path.join(tmp, "lib.so"))
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", os.path.join(tmp, "lib.so")], cwd=tmp, capture_output=True)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for f in sorted(os.listdir(tmp)):
    if "gfx950" not in f:
        continue
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if flt and flt not in name:
            continue
        rows.append((int(g("private_segment_fixed_size")), name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count"), blk.split()[0]))
rows.sort(key=lambda r: (-r[0], r[1]))
shutil.rmtree(tmp)
for r in rows:
    print(f"scratch={r[0]:5d} spill={r[5]:>3s} agpr={r[6]:>3s} vgpr={r[2]:>3s} sgpr={r[3]:>3s} lds={r[4]:>6s} {r[1][:150]}")
print(len(rows), "kernels")
