"""Developer sweep: the halo conv's time against the number of K slices (ST_HALO_BLOCKS = blocks aimed at; dev build
-DST_DEV_CONFIGS, ST_VARIANT=<name>).  One process per setting (the knob is read once).
usage: conv_splitk_sweep.py            -> runs itself for the step's three 3x3 levels and several targets"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
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
for shape in ((1, 1280, 32, 1280), (1, 640, 64, 640), (1, 320, 128, 320), (1, 2560, 32, 1280), (1, 1920, 32, 1280)):
    line = f"conv N={shape[0]} Cin={shape[1]} H={shape[2]} Cout={shape[3]}:"
    for target in (40, 80, 120, 160, 200, 240, 320):
        env = dict(os.environ, ST_HALO_BLOCKS=str(target))
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in shape], capture_output=True, text=True, env=env)
        r = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        line += f"  {target}: {r[0].split()[1] if r else 'ERR'}"
    print(line, flush=True)
