"""In-tree build of the gfx950 operator library with hipcc (no cmake, no JIT cache).

`python -m stabletriton_amd.build` or `build_library()` compiles every
csrc/*.hip to an object (in parallel) and links
stabletriton_amd/lib/libstabletriton_amd.so.  The .so is git-ignored but
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBNAME = "libstabletriton_amd.so"
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified register file), so the
# softmax / epilogue VALU code reads them without v_accvgpr_read/write copies.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the operator library cannot be built")
    return exe


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str) -> str:
    h = hashlib.sha256()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))       # every internal header
    for dep in [path] + headers + [os.path.join(PKG, "..", "include", "stabletriton_amd.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src[:-4] + ".o")
        stamp = obj + ".sha"
        dg = _digest(sp)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg)
        jobs.append((sp, obj, stamp, dg, fresh))

    def compile_one(job):
        sp, obj, stamp, dg, fresh = job
        if fresh:
            return obj
        cmd = [hipcc, *FLAGS, "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {sp}:\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(dg)
        return obj

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    out = lib_path()
    link_stamp = os.path.join(LIBDIR, "link.sha")
    want = _link_digest([j[3] for j in jobs])
    linked = os.path.exists(link_stamp) and open(link_stamp).read() == want
    if force or not os.path.exists(out) or not linked or any(not j[4] for j in jobs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        with open(link_stamp, "w") as f:
            f.write(want)
    return out


def _link_digest(source_digests) -> str:
    """What the linked library was made of: the digests of all its sources, in order (no file times: a copied tree keeps none)."""
    return hashlib.sha256("\n".join(source_digests).encode()).hexdigest()


def stale_sources() -> list:
    """Sources whose object (or the linked library) was not built from what is in the tree now: [] = the .so is this source.
    A failed compile leaves the previous .so in place; tests and bench.py refuse to measure that (tests/conftest.py)."""
    objdir = os.path.join(LIBDIR, "obj")
    out, digests = [], []
    for src in _sources():
        obj = os.path.join(objdir, src[:-4] + ".o")
        stamp = obj + ".sha"
        dg = _digest(os.path.join(CSRC, src))
        digests.append(dg)
        if not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg):
            out.append(src)
    link_stamp = os.path.join(LIBDIR, "link.sha")
    if not out and not (os.path.exists(lib_path()) and os.path.exists(link_stamp) and open(link_stamp).read() == _link_digest(digests)):
        out.append(LIBNAME + " (objects not linked)")
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
