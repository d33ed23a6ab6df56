"""Developer A/B of whole denoise steps on ONE box: the product library against tools/_variants/<name> builds (same ABI),
alternating processes so that device-to-device and thermal drift cancel.
usage: python tools/ab_step.py [--batch B] [--dtype bf16|fp16] [--rounds R] variantA variantB ...     ('product' = the in-tree build;
       'product:<pass>' = the same with the graph pass stabletriton_amd.optimization.<pass> switched off;
       '<variant>+KNOB=value' = a -DST_DEV_CONFIGS variant with that developer knob set;
       'product:nohints' / 'product:allhints' / 'product:hintcap<n>' = next-weights hints off / for every matrix / only for weight matrices of at most n MB)"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(variant, batch, dtype_name, latent, fp8=False):
    import torch
    from tools.devlib import use_variant
    variant, _, without = variant.partition(":")          # "product:fuse_skip_cat" = the build with that graph pass switched off
    use_variant(None if variant == "product" else variant)
    if without == "nohints" or without == "allhints" or without.startswith("hintcap"):
        # next-weights hints off / for every matrix (rounds 3-4) / only for weight matrices up to <n> MB ("product:hintcap8") instead of the rule of ops._next_weights
        from stabletriton_amd import ops as ops_mod
        cap = 0 if without == "nohints" else (1 << 40) if without == "allhints" else int(without[7:]) << 20
        ops_mod.HINT_MAX_BYTES = ops_mod.HINT_MAX_BYTES_SMALL_ROWS = ops_mod.HINT_MAX_BYTES_FP8 = cap
    elif without == "nolead" or (without.startswith("lead") and without[4:].isdigit()):      # strided touches of the large matrices off / 2^n lines per row
        from stabletriton_amd import ops as ops_mod
        ops_mod.HINT_LEAD_SHIFT = None if without == "nolead" else int(without[4:])
    elif without:
        import stabletriton_amd.optimization as opt_mod
        assert hasattr(opt_mod, without), f"no pass {without}"
        setattr(opt_mod, without, lambda gm, *a, **k: 0)
    from stabletriton_amd import synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.pipeline import DenoiseLoop
    from stabletriton_amd.scheduler import euler_discrete_tables
    from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
    dev = torch.device("cuda:0")
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype_name]
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE)
    m = m.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
    synth.fill_module_(m, 0)
    gm = optimize_model(m, cuda_graph=False, fp8=fp8)
    loop = DenoiseLoop(gm, batch, latent, dtype, dev, euler_discrete_tables(50), mode="step")
    x = synth.denoise_inputs(batch, latent, 1234, device=dev)
    loop.set_conditioning(x["encoder_hidden_states"].to(dtype), x["text_embeds"].to(dtype), x["time_ids"].to(dtype))
    loop.set_noise(x["latent"])
    with torch.no_grad():
        loop.capture()
        loop.run_steps(10)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            loop.run_steps(20)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print(f"RESULT {variant}{':' + without if without else ''} {best:.4f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--one")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--fp8", action="store_true", help="the projections of the fp8 plan on e4m3 operands (bf16 modules)")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.one:
        one(a.one, a.batch, a.dtype, a.latent, a.fp8)
        sys.exit(0)
    res = {v: [] for v in a.variants}
    for r in range(a.rounds):
        for v in a.variants:
            name, *sets = v.split("+")                   # "dev+ST_ATT_NW=4": the variant with that developer knob in the child's environment
            env = dict(os.environ, **dict(kv.split("=", 1) for kv in sets))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, "--batch", str(a.batch), "--dtype", a.dtype, "--latent", str(a.latent)] + (["--fp8"] if a.fp8 else []),
                                 capture_output=True, text=True, env=env)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(v, "FAILED", out.stderr[-800:])
                continue
            res[v].append(float(line[0].split()[2]))
            print(f"round {r} {v}: {res[v][-1]:.3f} ms/step", flush=True)
    for v, xs in res.items():
        if xs:
            print(f"{v:12s} batch {a.batch} {a.dtype}: min {min(xs):.3f}  mean {sum(xs) / len(xs):.3f} ms/step  ({a.batch * 1e3 / min(xs):.2f} it/s)")
