"""Developer sweep: the halo conv's time against the number of K slices (ST_HALO_SPLITS = slices; 0 = the launcher's own time
model; dev build -DST_DEV_CONFIGS, ST_VARIANT=<name>).  One process per setting (the knob is read once).
usage: conv_splitk_sweep.py [batch ...] [refiner]   -> runs itself for the step's 3x3 shapes at the given batch sizes (default 1 2 4)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 5:
    sys.path.insert(0, ROOT)
    import torch
    from tools.op_bench import timeit, rnd  # noqa: E402  (selects the ST_VARIANT build)
    from stabletriton_amd import ops
    N, Cin, H, Cout = (int(v) for v in sys.argv[1:5])
    cl = torch.channels_last
    x = rnd(N, Cin, H, H).contiguous(memory_format=cl)
    w = (rnd(Cout, Cin, 3, 3) * (Cin * 9) ** -0.5).contiguous(memory_format=cl)
    b = rnd(Cout)
    y = ops.conv2d(x, w, b, 1, 1)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    assert err < 2e-2, f"conv result is wrong: {err}"
    us = timeit(lambda: ops.conv2d(x, w, b, 1, 1))
    print(f"RESULT {us:.1f}")
    sys.exit(0)
refiner = "refiner" in sys.argv[1:]                        # SDXL-refiner widths (config #5) instead of SDXL-base's
batches = [int(v) for v in sys.argv[1:] if v != "refiner"] or [1, 2, 4]      # (a bare batch list: the self-invocations above pass four numbers)
for nb in batches:
    base = ((nb, 1280, 32, 1280), (nb, 640, 64, 640), (nb, 320, 128, 320), (nb, 2560, 32, 1280), (nb, 1920, 32, 1280), (nb, 1280, 64, 640))
    refi = ((nb, 1536, 32, 1536), (nb, 768, 64, 768), (nb, 384, 128, 384), (nb, 3072, 32, 1536), (nb, 1536, 64, 768), (nb, 768, 128, 384))
    for shape in (refi if refiner else base):
        line = f"conv N={shape[0]} Cin={shape[1]} H={shape[2]} Cout={shape[3]}:"
        for splits in (0, 1, 2, 3, 4, 5, 6, 8):
            env = dict(os.environ, ST_HALO_SPLITS=str(splits))
            out = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in shape], capture_output=True, text=True, env=env)
            r = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not r and splits == 0:
                print(out.stderr[-600:], flush=True)
            line += f"  {'model' if splits == 0 else splits}: {r[0].split()[1] if r else 'ERR'}"
        print(line, flush=True)
