"""Developer A/B: the halo conv's K split summed in-launch (last arriver of a tile) against the separate combine launch
(splitk_combine_kernel), per shape and number of slices; also checks that both give the same bits.  Needs a dev build
(-DST_DEV_CONFIGS: ST_HALO_SEP, ST_HALO_BLOCKS - the slice knob of the tree that experiment was made on), ST_VARIANT=<name>.  One process per setting (the knobs are read once).
usage: conv_sep_ab.py            -> the step's 3x3 shapes x {in-launch, separate} x several slice targets"""
import hashlib
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
    rb = rnd(N, Cout)
    y, st = ops.conv2d(x, w, b, 1, 1, rowbias=rb, emit_colstats=True)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1) + rb.float()[:, :, None, None]
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    assert err < 2e-2, f"conv result is wrong: {err}"
    if st is not None:      # the GroupNorm partials add up to the column sums of what was stored
        s = st.buf.double().view(-1, Cout, 2).sum(0).cpu()
        yy = y.float().permute(0, 2, 3, 1).reshape(-1, Cout).double().cpu()
        assert torch.allclose(s[:, 0], yy.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(s[:, 1], (yy * yy).sum(0), rtol=1e-4, atol=1e-2), "column statistics are wrong"
    us = timeit(lambda: ops.conv2d(x, w, b, 1, 1, rowbias=rb, emit_colstats=True))
    print(f"RESULT {us:.1f} {hashlib.sha1(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:10]} rows={st.rows if st is not None else 0}")
    sys.exit(0)
for shape in ((1, 1280, 32, 1280), (1, 2560, 32, 1280), (1, 1920, 32, 1280), (1, 640, 64, 640), (1, 1280, 64, 640), (1, 320, 128, 320), (1, 640, 128, 320),
              (4, 1280, 32, 1280), (4, 640, 64, 640)):
    print(f"conv N={shape[0]} Cin={shape[1]} H={shape[2]} Cout={shape[3]}:", flush=True)
    for target in (160, 240, 320, 480):
        line = f"   target {target:3d} blocks:"
        for sep in (0, 1):
            env = dict(os.environ, ST_HALO_BLOCKS=str(target), ST_HALO_SEP=str(sep))
            out = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in shape], capture_output=True, text=True, env=env)
            r = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            line += f"   {'separate ' if sep else 'in-launch'} {r[0][7:] if r else 'ERR ' + out.stderr[-200:]}"
        print(line, flush=True)
