"""Developer A/B: self-attention with the key-range split (two compute waves per 32-row group, each on half of the keys;
`attn32i_kernel<E, 6, true, 2>`) against the three-wave blocks it replaces, on a -DST_DEV_CONFIGS build of attention.hip
(ST_VARIANT=<name>, knob ST_ATT_KS = 0 / 1, read once per process).  Every case is checked against a float64 softmax of the
same 16-bit inputs before it is timed.  usage: ST_VARIANT=ks python tools/attn_ks_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 5:
    sys.path.insert(0, ROOT)
    import torch
    from tools.op_bench import timeit, rnd
    from stabletriton_amd import ops
    B, T, H = (int(v) for v in sys.argv[1:4])
    gain = float(sys.argv[4])
    torch.manual_seed(0)
    q, k, v = rnd(B, T, H * 64) * gain, rnd(B, T, H * 64) * gain, rnd(B, T, H * 64)
    out = ops.attention(q, k, v, H, 0.125)
    qd, kd, vd = (t.double().view(B, T, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * 0.125, -1) @ vd).transpose(1, 2).reshape(B, T, H * 64)
    err = (out.double() - ref).abs().max().item()
    print(f"RESULT {timeit(lambda: ops.attention(q, k, v, H, 0.125)):.2f} {err:.3e} {ref.abs().max().item():.3f}")
    sys.exit(0)
for B, T, H, gain in ((1, 1024, 20, 1.0), (1, 1024, 20, 6.0), (1, 1024, 24, 1.0), (1, 896, 20, 1.0), (1, 1000, 20, 1.0), (2, 1024, 20, 1.0), (1, 4096, 10, 1.0)):
    line = f"B={B} T={T} H={H} gain={gain}:"
    for ks in (0, 1):
        env = dict(os.environ, ST_ATT_KS=str(ks))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), str(T), str(H), str(gain)], capture_output=True, text=True, env=env)
        r = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        line += f"  ks={ks}: " + (" ".join(r[0].split()[1:]) if r else "ERR " + out.stderr[-300:])
    print(line, flush=True)
