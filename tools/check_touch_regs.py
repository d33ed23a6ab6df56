"""Static check of the fire-and-forget weight / operand touches (csrc: touch_next_weights, the epilogue-operand touches of the GEMM
prologues): they are inline-asm `global_load_dword vN, ...` whose destination the compiler believes valid at once, so under
register pressure it may split vN's live range (copy it away) and hand vN to something else while the loads are still in flight -
the late return then overwrites a live value (round 5: an address register of the new register epilogue -> memory aperture
violation in the denoise step).  For every such load this scans the emitted assembly up to the asm `s_waitcnt vmcnt(..)` that
retires it and reports any other instruction that writes vN in between.

    python tools/check_touch_regs.py [source.hip ...]      (default: every csrc/*.hip; compiles each with -S, minutes of CPU)
Exit status 1 if a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stabletriton_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "--cuda-device-only", "-S"]


def emit(src, out):
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", out], check=True, capture_output=True)
    return out


def dest_regs(ins):
    """registers an instruction writes (first operand), as a set of VGPR numbers"""
    m = re.match(r"\s*(\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b", ins)
    if not m or m.group(1).startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "s_", ";", "flat_store", "v_cmp", "v_cmpx", "global_atomic")):
        return set()
    if m.group(1).startswith("ds_read") or m.group(1).startswith(("v_", "global_load", "buffer_load", "scratch_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
        if m.group(5) is not None:
            return {int(m.group(5))}
        return set(range(int(m.group(3)), int(m.group(4)) + 1))
    return set()


def check(asm_path):
    bad = []
    kernel = None
    lines = open(asm_path).read().split("\n")
    in_asm = False
    pending = {}          # vN -> (kernel, line number of the touch)
    saved = {}            # label -> pending set at an unconditional forward branch to it (the code behind `s_branch` is another path)
    else_stack = []
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\S+:", ln):
            kernel = ln.rstrip(":")
            pending, saved, else_stack = {}, {}, []
        m_lab = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m_lab and m_lab.group(1) in saved:
            pending = {**pending, **saved.pop(m_lab.group(1))}
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        if in_asm:
            m = re.match(r"global_load_dword v(\d+), v\[\d+:\d+\], off\s*$", s)
            if m:
                pending[int(m.group(1))] = (kernel, i + 1)
                continue
            if s.startswith("s_waitcnt vmcnt("):
                pending = {}
            continue
        if s.startswith("s_endpgm"):
            pending = {}
            continue
        if re.match(r"s_xor_b64 exec, exec, ", s):      # the ELSE lanes of a predicated diamond: other lanes than the ones that touched
            else_stack.append(pending)
            pending = {}
            continue
        if else_stack and re.match(r"s_or_b64 exec, exec, ", s):      # rejoin
            pending = {**else_stack.pop(), **pending}
            continue
        m_br = re.match(r"s_branch (\.LBB\d+_\d+)", s)
        if m_br:             # what follows in the file is reached from elsewhere: a linear scan must not carry this path's touches into it
            saved[m_br.group(1)] = {**saved.get(m_br.group(1), {}), **pending}
            pending = {}
            continue
        if pending:
            for r in dest_regs(s) & set(pending):
                bad.append((asm_path, kernel, pending[r][1], i + 1, s))
    return bad


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    tmp = tempfile.mkdtemp()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        outs = list(ex.map(lambda s: emit(s, os.path.join(tmp, os.path.basename(s)[:-4] + ".s")), srcs))
    bad = [b for o in outs for b in check(o)]
    for path, kernel, l0, l1, ins in bad:
        print(f"{os.path.basename(path)}: {kernel}: touch at line {l0}, destination rewritten at line {l1}: {ins}")
    print(f"{len(outs)} translation units, {len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
