"""Developer timing of one Linear shape: product dispatch, forced configurations, hipBLASLt (torch) - warm and cold weights.
usage: gemm_one.py M K N [g]   (needs ST_VARIANT=dev for the forced rows)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import _C, ops
from tools.op_bench import timeit, rnd
lib = _C.load()
force = getattr(lib, "st_debug_force_gemm", None)
if force is not None:
    force.argtypes, force.restype = [ctypes.c_int, ctypes.c_int], None
M, K, N = (int(v) for v in sys.argv[1:4])
geglu = len(sys.argv) > 4 and "g" in sys.argv[4]
rows = 2 * N if geglu else N
x, b = rnd(M, K), rnd(rows)
fl = 2.0 * M * K * rows
for ncopy, tag in ((1, "warm"), (max(1, min(32, int(600e6 // (rows * K * 2)))), "cold")):
    ws = [rnd(rows, K) * K ** -0.5 for _ in range(ncopy)]
    it = [0]
    def ours():
        it[0] += 1
        return ops.linear(x, ws[it[0] % ncopy], b, geglu=geglu)
    def vendor():
        it[0] += 1
        return torch.nn.functional.linear(x, ws[it[0] % ncopy], b)
    line = f"{tag} weights ({ncopy} copies): "
    if force is not None: force(-1, -1)
    us = timeit(ours, iters=max(20, ncopy)); line += f"product {us:.1f} us ({fl/us/1e6:.0f} TF/s)"
    if force is not None:
        for cfg, name in ((100, "8p"), (101, "8p160"), (19, "256x128"), (9, "128x128"), (28, "128x160"), (10, "64x128"), (8, "128x64"), (27, "128x80"), (26, "64x80 W4"), (15, "64x128 U2"), (16, "128x64 U2"), (11, "64x128 S6"), (12, "128x64 S6"),
                          (5, "128x128 N4S3"), (29, "128x128 N4S2"), (30, "128x64 N4S3"), (31, "64x128 N4S3"), (3, "128x64 N4S4"), (6, "64x64 N4S3"), (32, "256x160 S2")):
            if geglu and cfg in (27, 26, 30, 6, 3):          # (odd n-tiles per wave: no value/gate pairing; the dispatch never picks it)
                continue
            force(cfg, 1)
            try:
                us = timeit(ours, iters=max(20, ncopy)); line += f" | {name} {us:.1f}"
            except Exception as e:
                line += f" | {name} n/a"
        force(-1, -1)
    if not geglu:
        us = timeit(vendor, iters=max(20, ncopy)); line += f" | hipBLASLt {us:.1f} ({fl/us/1e6:.0f} TF/s)"
    print(line, flush=True)
