#!/usr/bin/env python3
"""Headline benchmark: denoise it/s of the SDXL-base UNet at 1024x1024 (latent
128x128), bs=1 per GPU, bf16, Euler-discrete loop captured as a hipGraph.

    python bench.py --gpus N --steps K --warmup W

A "step" is one denoise iteration (UNet forward + scheduler update) of one
prompt per GPU.  N>1 is launched by torch.distributed.run (one rank per GPU):
rank 0 generates the synthetic weights and broadcasts them over RCCL, then every
rank runs its own independent trajectory (no per-step traffic, weak scaling).
Rank 0 prints ONE JSON line.  At N=1 the line also carries a per-kernel roofline
(HIP-event census of one eager step) and the CPU baseline (the oracle restatement
of the reference's eager path, timed on the host cores for a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stabletriton_amd import ops, parallel, synth                      # noqa: E402
from stabletriton_amd.optimization import optimize_model              # noqa: E402
from stabletriton_amd.pipeline import DenoiseLoop                     # noqa: E402
from stabletriton_amd.scheduler import euler_discrete_tables          # noqa: E402
from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel     # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (guide: ~2.5 PF)
PEAK_HBM_GBS = 8000.0          # HBM3E spec
BOUND = {"linear": "mfma", "conv2d": "mfma", "attention_self": "mfma", "attention_cross": "hbm",
         "group_norm": "hbm", "layer_norm": "hbm", "geglu": "hbm"}
KERNEL = {"linear": "gemm_dma_kernel<bf16,...,CONV=false>", "conv2d": "gemm_dma_kernel<bf16,...,CONV=true>",
          "attention_self": "attn_bf16_kernel / attn16_bf16_kernel", "attention_cross": "attn16_bf16_kernel",
          "group_norm": "gn_stats_nhwc+gn_finalize+gn_apply_nhwc", "layer_norm": "ln_kernel", "geglu": "geglu_kernel"}


def build_model(dev, dtype, rank, world):
    with torch.device("meta"):
        model = UNet2DConditionModel(SDXL_BASE)
    model = model.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
    t0 = time.time()
    if rank == 0:
        synth.fill_module_(model, 0)
    n_bcast = parallel.broadcast_module(model, src=0) if world > 1 else 0
    torch.cuda.synchronize(dev)
    return model, time.time() - t0, n_bcast


def census(loop):
    """One eager step with HIP events around every operator launch (same stream).

    The host needs ~20-40 us of Python per launch, more than most kernels run, so a naive eager
    pass would time an idle GPU waiting for the host.  The measured step is therefore enqueued
    behind a spin kernel: by the time the GPU reaches it every launch and event record is already
    in the queue, kernels run back to back as in the graph, and event deltas are device times."""
    store = []
    loop_mode, loop.mode = loop.mode, "eager"
    ops.set_census(None)
    loop.run_steps(1)                      # warm the eager path
    torch.cuda.synchronize()
    step0 = int(loop.step.item())
    torch.cuda._sleep(int(6e8))            # ~0.25 s head start for the host
    # cost of the event pair itself (nothing launched in between), measured in the same queued regime
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in empty:
        a.record(); b.record()
    ops.set_census(store)
    for i in range(1):
        loop._step_const((step0 + i) % loop.n_steps)
    ops.set_census(None)
    torch.cuda.synchronize()
    loop.mode = loop_mode
    pair_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    fam = {}
    for name, flops, nbytes, e0, e1, _tag in store:
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        f["launches"] += 1
        f["ms"] += max(e0.elapsed_time(e1) - pair_ms, 0.0)
        f["flops"] += flops
        f["bytes"] += nbytes
    if os.environ.get("ST_CENSUS_SHAPES"):           # developer view: time per (operator, shape)
        rows = {}
        for name, flops, nbytes, e0, e1, tag in store:
            r = rows.setdefault((name, tag), [0, 0.0, 0.0])
            r[0] += 1; r[1] += max(e0.elapsed_time(e1) - pair_ms, 0.0); r[2] += flops
        for (name, tag), (cnt, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            print(f"  {ms:7.3f} ms  x{cnt:3d}  {ms / cnt * 1e3:7.1f} us  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {name:16s} {tag}", file=sys.stderr)
    return fam


def measured_traffic(name):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json), or None."""
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        return json.load(open(latest)).get(f"{name}_bytes_per_launch")
    except Exception:
        return None


def roofline_of(name, f):
    sec = f["ms"] * 1e-3
    if BOUND[name] == "mfma":
        ach, peak, unit = f["flops"] / sec / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s"
    else:
        ach, peak, unit = f["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
    return {"kernel": KERNEL[name], "op": name, "bound": BOUND[name], "achieved": round(ach, 2), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4), "traffic": measured_traffic(name), "launches_per_step": f["launches"],
            "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2), "ms_per_step": round(f["ms"], 3)}


def cpu_baseline(model, latent_hw):
    """Oracle (CPU restatement of the reference eager path, fp32) on the host cores."""
    from oracle import unet_oracle as orc          # checker/baseline only, never on the product path
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    x = synth.denoise_inputs(1, latent_hw, 1234)
    cores = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.time()
        orc.unet_forward(sd, x["latent"], torch.tensor(981.0), x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
        dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"1 UNet step, latent {latent_hw}x{latent_hw}, bs=1, fp32 eager torch on {cores} threads, "
                      f"first call (no warm-up), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--mode", choices=["auto", "loop", "step", "eager"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-census", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a same-device rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    rank, world, local = parallel.init_from_env(args.backend)
    if args.same_device:
        local = 0
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16
    n_sched = 50

    model, t_weights, n_bcast = build_model(dev, dtype, rank, world)
    gm = optimize_model(model, cuda_graph=False)
    mode = args.mode
    if mode == "auto":
        mode = "loop" if (args.steps % n_sched == 0 and args.warmup % n_sched == 0) else "step"
    loop = DenoiseLoop(gm, args.batch, args.latent, dtype, dev, euler_discrete_tables(n_sched), mode=mode)
    x = synth.denoise_inputs(args.batch, args.latent, 1234 + rank, device=dev)
    loop.set_conditioning(x["encoder_hidden_states"].to(dtype), x["text_embeds"].to(dtype), x["time_ids"].to(dtype))
    loop.set_noise(x["latent"])

    with torch.no_grad():
        t0 = time.time()
        loop.capture()
        t_capture = time.time() - t0
        if args.warmup:
            loop.run_steps(args.warmup)
        torch.cuda.synchronize(dev)
        parallel.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loop.run_steps(args.steps)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        parallel.barrier()
        elapsed = parallel.max_over_ranks(elapsed, dev)
        finite = bool(torch.isfinite(loop.latent).all())

        result = {
            "metric": "denoise it/s, SDXL UNet 1024x1024 50-step, bs=1 per GPU",
            "value": round(world * args.batch * args.steps / elapsed, 3),
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SDXL-base UNet, latent {args.latent}x{args.latent} (1024x1024 px), bs={args.batch}/GPU, "
                                   f"{n_sched}-step Euler-discrete loop, hipGraph mode={mode}, no CFG",
                       "parallelism": f"prompt-parallel x{world}", "weights": "synthetic seed 0",
                       "weight_broadcasts": n_bcast},
            "finite": finite, "capture_s": round(t_capture, 2), "weights_s": round(t_weights, 2),
        }
        if rank == 0 and world == 1:
            if not args.no_census:
                fam = census(loop)
                roofs = {k: roofline_of(k, v) for k, v in fam.items()}
                dominant = max(fam, key=lambda k: fam[k]["ms"])
                result["roofline"] = roofs[dominant]
                result["kernels"] = sorted(roofs.values(), key=lambda r: -r["ms_per_step"])
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(model, args.latent)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
