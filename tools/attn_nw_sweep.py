"""Developer sweep: self-attention block shapes (compute waves per block, ST_ATT_NW; -DST_DEV_CONFIGS build of attention.hip,
ST_VARIANT=<name>) against the launch rule's own choice, for the step's two token counts at batch 1 / 2 / 4.
One process per setting (the knob is read once).  usage: attn_nw_sweep.py [refiner]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 4:
    sys.path.insert(0, ROOT)
    import torch
    from tools.op_bench import timeit, rnd
    from stabletriton_amd import ops
    B, T, H = (int(v) for v in sys.argv[1:4])
    q, k, v = rnd(B, T, H * 64), rnd(B, T, H * 64), rnd(B, T, H * 64)
    print(f"RESULT {timeit(lambda: ops.attention(q, k, v, H, 0.125)):.1f}")
    sys.exit(0)
refiner = "refiner" in sys.argv[1:]             # SDXL-refiner's heads (24 at 1,024 tokens, 12 at 4,096) instead of SDXL-base's
for B in (1, 2, 4):
    for T, H in (((1024, 24), (4096, 12)) if refiner else ((1024, 20), (4096, 10))):
        line = f"B={B} T={T} H={H}:"
        for nw in (0, 3, 4, 5, 6, 7, 8):
            env = dict(os.environ, ST_ATT_NW=str(nw))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), str(T), str(H)], capture_output=True, text=True, env=env)
            r = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            line += f"  {'rule' if nw == 0 else nw}: {r[0].split()[1] if r else 'ERR'}"
        print(line, flush=True)
