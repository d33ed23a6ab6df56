#!/usr/bin/env python3
"""Headline benchmark: denoise it/s of the SDXL-base UNet at 1024x1024 (latent
128x128), bs=1 per GPU, bf16, Euler-discrete loop captured as a hipGraph.

    python bench.py --gpus N --steps K --warmup W

A "step" is one denoise iteration (UNet forward + scheduler update) of one
prompt per GPU.  N>1 runs one rank per GPU: under torch.distributed.run the
ranks come from the environment; invoked plainly, this process spawns the N
ranks itself (fresh child processes, before anything touches the GPU) and
relays rank 0's line.  Rank 0 generates the synthetic weights and broadcasts
them over RCCL, then every rank runs its own independent trajectory (no
per-step traffic, weak scaling).  Rank 0 prints ONE JSON line.  At N=1 the line
also carries a per-kernel roofline (HIP-event census of one eager step scaled to
the graph-replay step time) and the CPU baseline (the oracle restatement of the
reference's eager path, timed on the host cores for a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 / fp16 MFMA (guide: ~2.5 PF)
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 through the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (guide: ~5 PF)
PEAK_HBM_GBS = 8000.0          # HBM3E spec
BOUND = {"linear": "mfma", "linear_xattn": "mfma", "linear_fp8": "mfma", "quantize_fp8": "hbm", "conv2d": "mfma", "attention_self": "mfma", "attention_cross": "hbm",
         "group_norm": "hbm", "layer_norm": "hbm", "geglu": "hbm", "split_f32": "hbm"}
KERNEL = {"split_f32": "split_rows_kernel (strict mode: fp32 -> split fp16 pair image of a matrix operand)", "linear_fp8": "gemm_dma_kernel<f8, CONV=false> (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3)", "quantize_fp8": "quant_fp8_kernel",
          "linear": "gemm_dma_kernel / gemm8p_kernel <bf16, CONV=false>",
          "linear_xattn": "gemm_dma_kernel<bf16, 128, 64, ..., XA=true> (query projection + text-context attention in its epilogue)", "conv2d": "conv_halo_kernel / gemm_dma_kernel<bf16, CONV=true>",
          "attention_self": "attn32i_kernel<7, true> / attn32i_kernel<4, true>", "attention_cross": "attn16v2_kernel<4, 1>",
          "group_norm": "gn_cols_finalize+gn_apply_nhwc (statistics from the producer's epilogue; standalone: gn_stats_nhwc+gn_finalize+gn_apply_nhwc)", "layer_norm": "ln_kernel", "geglu": "geglu_kernel"}


STRICT_KERNEL = {"linear": "gemm_dma_kernel<fsp, CONV=false> (split fp32 operands: 3 x v_mfma_f32_16x16x32_f16 per 32 k)",
                 "conv2d": "conv_halo_kernel<fsp> / gemm_dma_kernel<fsp, CONV=true> (split operands)",
                 "attention_self": "attn_split_kernel<4, PRE>", "attention_cross": "attn_split_kernel<4, false>"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--latent-w", type=int, default=None, help="latent width when it differs from --latent (the height): an aspect bucket, "
                    "e.g. --latent 152 --latent-w 104 = 1216 x 832 px; a side line, the headline workload is square")
    ap.add_argument("--mode", choices=["auto", "loop", "step", "eager"], default="auto")
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16",
                    help="compute / storage type of the timed run (same matrix-pipe rate; fp16 is the reference call site's own type)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements of the N=1 line (loop-graph replay when the "
                                                             "timed mode is 'step', the other 16-bit type, strict fp32)")
    ap.add_argument("--fp8", action="store_true", help="transformer-block projections on the fp8 matrix pipe (BASELINE config #5 mode; "
                                                       "a separate line with dtype fp8, never the headline)")
    ap.add_argument("--model", choices=["base", "refiner"], default="base",
                    help="SDXL-base (the headline, BASELINE configs #2-#4) or the SDXL-refiner UNet (config #5; parity unpinned: the reference has no refiner)")
    ap.add_argument("--img2img", type=float, default=None, metavar="STRENGTH",
                    help="img2img start (the refiner's use): the trajectory starts at schedule entry n - int(n * STRENGTH) from init latent + noise; "
                         "the timed steps are denoise steps from there (step graph, wrapping round the schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-census", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a same-device rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# plain `python bench.py --gpus N`: spawn the ranks (nothing in this process has touched the GPU yet)
# ---------------------------------------------------------------------------------------------------
def self_launch(args, cmd=None) -> int:
    """Spawn `args.gpus` ranks of `cmd` (default: this script with the same arguments), relay rank 0's stdout,
    return non-zero if any rank failed."""
    if cmd is None:
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="st_bench_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = open(os.path.join(logdir, f"rank{r}.out"), "w")
        err = open(os.path.join(logdir, f"rank{r}.err"), "w")
        procs.append((subprocess.Popen(cmd, env=env, stdout=out, stderr=err), out, err))
    rc = 0
    deadline = time.time() + 3000
    while True:                          # a rank that dies leaves the others stuck in the rendezvous or a collective: stop at the first failure
        codes = [p.poll() for p, _, _ in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = -9
            break
        time.sleep(0.2)
    for _, out, err in procs:
        out.close(); err.close()
    if rc != 0:
        for p, _, _ in procs:          # a failed rank leaves the others stuck in a collective: end exactly our children
            if p.poll() is None:
                p.kill()
        for r in range(args.gpus):
            tail = open(os.path.join(logdir, f"rank{r}.err")).read()[-2000:]
            sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{tail}\n")
        return rc if rc > 0 else 1
    sys.stdout.write(open(os.path.join(logdir, "rank0.out")).read())
    sys.stdout.flush()
    sys.stderr.write(open(os.path.join(logdir, "rank0.err")).read()[-4000:])
    return 0


def model_spec(name):
    from stabletriton_amd.unet import SDXL_BASE, SDXL_REFINER
    return SDXL_REFINER if name == "refiner" else SDXL_BASE


def build_model(dev, dtype, rank, world, spec):
    import torch
    from stabletriton_amd import parallel, synth
    from stabletriton_amd.unet import UNet2DConditionModel
    with torch.device("meta"):
        model = UNet2DConditionModel(spec)
    model = model.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
    t0 = time.time()
    if rank == 0:
        synth.fill_module_(model, 0)
    torch.cuda.synchronize(dev)
    t_fill = time.time() - t0
    t0 = time.time()
    n_bcast = parallel.broadcast_module(model, src=0) if world > 1 else 0
    torch.cuda.synchronize(dev)
    return model, t_fill, time.time() - t0, n_bcast


def census(loop):
    """One eager step with HIP events around every operator launch (same stream).

    The host needs ~20-40 us of Python per launch, more than most kernels run, so a naive eager
    pass would time an idle GPU waiting for the host.  The measured step is therefore enqueued
    behind a spin kernel: by the time the GPU reaches it every launch and event record is already
    in the queue, kernels run back to back as in the graph, and event deltas are device times.

    Returns (families, step_ms, marker_ms, glue_ms): per family the summed event deltas with the cost of one
    event record removed (kernel-only time), the step's kernel-only time including the torch glue kernels
    between the operator launches, the per-record cost and the glue time."""
    import torch
    from stabletriton_amd import _C
    from tools.census import Census        # wraps the C-ABI entry points of the loaded library (nothing in the product knows)
    loop_mode, loop.mode = loop.mode, "eager"
    loop.run_steps(1)                      # warm the eager path
    torch.cuda.synchronize()
    step0 = int(loop.step.item())
    torch.cuda._sleep(int(6e8))            # ~0.25 s head start for the host
    # cost of an event record (nothing launched in between), measured in the same queued regime
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in empty:
        a.record(); b.record()
    with Census(_C.load()) as c:
        loop._step_const(step0 % loop.n_steps)
    store = c.records
    torch.cuda.synchronize()
    loop.mode = loop_mode
    # cost of one event record: between two operator launches with no torch kernel in between, the gap e1(i) -> e0(i+1)
    # is exactly one record, so the median gap is that cost in the regime it is paid in (the empty pairs bound it from below)
    gaps = sorted(store[i][4].elapsed_time(store[i + 1][3]) for i in range(len(store) - 1))
    pair_ms = max(gaps[len(gaps) // 2], sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2])
    fam = {}
    for name, flops, nbytes, e0, e1, _tag in store:
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        f["launches"] += 1
        f["ms"] += max(e0.elapsed_time(e1) - pair_ms, 0.0)
        f["flops"] += flops
        f["bytes"] += nbytes
    glue_ms = sum(max(g - pair_ms, 0.0) for g in gaps)           # torch kernels between operator launches (cat, casts)
    step_ms = sum(f["ms"] for f in fam.values()) + glue_ms
    if os.environ.get("ST_CENSUS_SHAPES"):           # developer view: time per (operator, shape)
        rows = {}
        for name, flops, nbytes, e0, e1, tag in store:
            r = rows.setdefault((name, tag), [0, 0.0, 0.0])
            r[0] += 1; r[1] += max(e0.elapsed_time(e1) - pair_ms, 0.0); r[2] += flops
        for (name, tag), (cnt, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            print(f"  {ms:7.3f} ms  x{cnt:3d}  {ms / cnt * 1e3:7.1f} us  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {name:16s} {tag}", file=sys.stderr)
        print(f"  census step {step_ms:.3f} ms (operators {sum(f['ms'] for f in fam.values()):.3f} ms, {len(store)} launches, "
              f"event record {pair_ms * 1e3:.2f} us)", file=sys.stderr)
    return fam, step_ms, pair_ms, glue_ms


def committed_profile(name, tag=""):
    """Per-family numbers of the committed rocprofv3 run of this command (profiles/rNN_families.json, written by
    tools/profile_families.py from the kernel-trace CSV) and the PMC traffic (profiles/rNN_traffic.json), or None."""
    import glob
    out = {}
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{tag}_families.json")))[-1]      # (the bf16 headline's: not r04_strict_* / r04_refiner_*)
        rec = json.load(open(latest)).get("families", {}).get(name)
        if rec:
            out = {"source": os.path.relpath(latest, ROOT), "ms_per_step": rec["ms_per_step"], "avg_launch_us": rec["avg_launch_us"],
                   "launches_per_step": rec["launches_per_step"]}
    except Exception:
        pass
    traffic = mfma_busy = None
    try:
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{tag}_traffic.json")))[-1]
        t = json.load(open(latest))
        traffic = t.get(f"{name}_bytes_per_launch")
        mfma_busy = t.get(f"{name}_mfma_busy")
        if out and t.get(f"{name}_flops_over_peak") is not None:      # useful flops under the counters (the attention's ones-block MFMAs are busy cycles, not work)
            out["pmc_flops_over_peak"] = t[f"{name}_flops_over_peak"]
    except Exception:
        pass
    return out, traffic, mfma_busy


def roofline_of(name, f, boundary_ms, committed=True, tag=""):
    """`boundary_ms`: what one launch costs in the replayed graph on top of its kernel-only time - the dependent-launch
    boundary in front of every kernel, which rocprofv3 attributes to the kernel (its trace shows back-to-back kernels with
    no gaps) and an event-bracketed launch does not see: (graph step time - census kernel-only time) / launches."""
    ms = f["ms"] + f["launches"] * boundary_ms
    sec = ms * 1e-3
    if BOUND[name] == "mfma":
        # (fp8 projections issue the block-scaled K=128 instruction: priced against the fp8 peak)
        ach, peak, unit = f["flops"] / sec / 1e12, (PEAK_FP8_TFLOPS if name == "linear_fp8" else PEAK_BF16_TFLOPS), "TFLOP/s"
    else:
        ach, peak, unit = f["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
    prof, traffic, mfma_busy = committed_profile(name, tag) if committed else ({}, None, None)
    r = {"kernel": KERNEL[name], "op": name, "bound": BOUND[name], "achieved": round(ach, 2), "peak": peak, "unit": unit,
         "frac": round(ach / peak, 4), "traffic": traffic, "launches_per_step": f["launches"],
         "avg_launch_us": round(ms * 1e3 / f["launches"], 2), "ms_per_step": round(ms, 3),
         "kernel_only_ms": round(f["ms"], 3)}
    if BOUND[name] == "mfma":
        r["flops_over_peak"] = r["frac"]          # algorithmic (useful) flops over the dense peak, live: what mfma_busy must be read against
    if mfma_busy is not None:
        r["mfma_busy"] = mfma_busy
    if prof:
        work = f["flops"] / 1e12 if BOUND[name] == "mfma" else f["bytes"] / 1e9
        prof["frac"] = round(work / (prof["ms_per_step"] * 1e-3) / peak, 4)
        r["rocprof"] = prof
    return r


def _time_loop(loop, steps, warmup, dev):
    import torch
    loop.capture()
    if warmup:
        loop.run_steps(warmup)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loop.run_steps(steps)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def extras(args, model, gm, loop, mode, dtype, dev, n_sched, cond, x):
    """Side measurements printed beside the headline (never `value`): the same build in the other modes a reader asks about.
      loop_graph_it_per_s   - one replay of the 50-step loop hipGraph (north_star's form) when the timed run was step mode
                              (the driver's --steps 20 --warmup 5 is not a whole number of loops);
      other16_it_per_s      - the other 16-bit type (fp16 <-> bf16), same protocol;
      strict_fp32_it_per_s  - the strict mode (fp32 storage, matrix operands as split fp16 pairs on the 16-bit MFMA, fp32
                              accumulation / softmax / statistics): the one that meets north_star's 1e-3 abs gate; `strict` holds its
                              step time and its own per-family roofline.
    A leg that fails is recorded under `extras_error`; the headline line is printed regardless."""
    import torch
    from stabletriton_amd import synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.pipeline import DenoiseLoop
    from stabletriton_amd.scheduler import euler_discrete_tables
    from stabletriton_amd.unet import UNet2DConditionModel
    spec = model_spec(args.model)
    out = {}

    def make(g, dt, md):
        lp = DenoiseLoop(g, args.batch, latent_hw(args), dt, dev, euler_discrete_tables(n_sched), cross_dim=spec.cross_dim, pooled_dim=spec.pooled_dim,
                         mode=md, n_time_ids=spec.n_time_ids)
        lp.set_conditioning(*(c.to(dt) for c in cond))
        lp.set_noise(x["latent"])
        return lp

    def leg(name, fn):
        """a side measurement must never cost the headline line: a failure (an OOM on a shared card, a broken side path) is recorded"""
        try:
            with torch.no_grad():
                fn()
        except Exception as e:          # noqa: BLE001
            out.setdefault("extras_error", {})[name] = f"{type(e).__name__}: {e}"[:300]
        torch.cuda.empty_cache()

    def loop_graph():
        lp = make(gm, dtype, "loop")
        ms = _time_loop(lp, n_sched, n_sched, dev)
        out["loop_graph_it_per_s"] = round(args.batch * 1e3 / ms, 3)

    def other16():
        other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
        with torch.device("meta"):
            m2 = UNet2DConditionModel(spec)
        m2 = m2.to_empty(device=dev).to(other).eval().requires_grad_(False)
        m2.load_state_dict({k: v.to(other) for k, v in model.state_dict().items()})
        lp = make(optimize_model(m2, cuda_graph=False), other, "step")
        ms = _time_loop(lp, 20, 5, dev)
        out["other16_it_per_s"] = {"dtype": "fp16" if other == torch.float16 else "bf16", "value": round(args.batch * 1e3 / ms, 3),
                                   "finite": bool(torch.isfinite(lp.latent).all())}

    def strict():
        with torch.device("meta"):
            m3 = UNet2DConditionModel(spec)
        m3 = m3.to_empty(device=dev).float().eval().requires_grad_(False)
        synth.fill_module_(m3, 0)
        lp = make(optimize_model(m3, cuda_graph=False), torch.float32, "step")
        ms = _time_loop(lp, 20, 5, dev)
        out["strict_fp32_it_per_s"] = round(args.batch * 1e3 / ms, 3)
        rec = {"ms_per_step": round(ms, 3), "finite": bool(torch.isfinite(lp.latent).all()),
               "arithmetic": "fp32 storage; Linear / conv / attention operands as split fp16 pairs (x ~ hi + lo * 2^-11), three "
                             "v_mfma_f32_16x16x32_f16 per product, fp32 accumulation, fp32 softmax and statistics"}
        if not args.no_census:          # the strict step's own families against the 16-bit MFMA peak (algorithmic flops: the 3x of the split is overhead)
            fam, census_ms, _, _ = census(lp)
            n_launch = sum(f["launches"] for f in fam.values())
            boundary = max(ms - census_ms, 0.0) / n_launch
            strict_prof = args.batch == 1 and args.model == "base" and args.img2img is None      # (profiles/rNN_strict_* is the bs=1 SDXL-base strict step)
            roofs = {k: roofline_of(k, v, boundary, committed=strict_prof, tag="_strict") for k, v in fam.items()}
            for k, r in roofs.items():
                r["kernel"] = STRICT_KERNEL.get(k, r["kernel"])
            rec["roofline"] = roofs[max(fam, key=lambda k: fam[k]["ms"])]
            rec["kernels"] = [{k: r[k] for k in ("op", "bound", "achieved", "unit", "frac", "launches_per_step", "ms_per_step", "kernel_only_ms")}
                              for r in sorted(roofs.values(), key=lambda r: -r["ms_per_step"])]
        out["strict"] = rec

    if mode == "step" and not args.fp8:
        leg("loop_graph", loop_graph)
    if not args.fp8 and dtype != torch.float32:
        leg("other16", other16)
        leg("strict", strict)
    return out


def host_cores():
    """(threads to use, description): physical cores of the host, capped by this process's affinity mask and cgroup CPU quota."""
    logical = os.cpu_count() or 1
    phys = None
    try:
        seen = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    seen.add((pid, cid))
                pid = cid = None
        phys = len(seen) or None
    except OSError:
        pass
    use = phys or logical
    try:
        use = min(use, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
            use = min(use, quota)
    except (OSError, ValueError):
        pass
    return max(1, use), f"{phys or '?'} physical / {logical} logical cores" + (f", cgroup quota {quota}" if quota else "")


def latent_hw(args):
    return args.latent if args.latent_w in (None, args.latent) else (args.latent, args.latent_w)


def cpu_baseline(model, latent_hw, spec):
    """Oracle (CPU restatement of the reference eager path, fp32) on the host cores: 1 warm-up + 2 timed steps, median."""
    import torch
    from oracle import unet_oracle as orc          # checker/baseline only, never on the product path
    from stabletriton_amd import synth
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    x = synth.denoise_inputs(1, latent_hw, 1234, cross_dim=spec.cross_dim, pooled_dim=spec.pooled_dim, n_time_ids=spec.n_time_ids)
    cores, desc = host_cores()
    before = torch.get_num_threads()
    torch.set_num_threads(cores)
    times = []
    try:
        with torch.no_grad():
            for _ in range(3):
                t0 = time.time()
                orc.unet_forward(sd, x["latent"], torch.tensor(981.0), x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
                times.append(time.time() - t0)
    finally:
        torch.set_num_threads(before)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(1.0 / dt, 5), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"UNet steps at latent {'x'.join(str(v) for v in synth.latent_size(latent_hw))}, bs=1, fp32 eager torch (oracle) on {cores} threads ({desc}): "
                      f"1 warm-up ({times[0]:.1f} s) + 2 timed ({times[1]:.1f}, {times[2]:.1f} s), median {dt:.1f} s per step"}


def rccl_report(path):
    """What RCCL logged at communicator init (NCCL_DEBUG=INFO to a per-rank file): rank count and transports seen."""
    try:
        txt = open(path, errors="replace").read()
    except OSError:
        return None
    import re
    nranks = sorted({int(m) for m in re.findall(r"nranks (\d+)", txt)})
    via = sorted(set(re.findall(r"via ([A-Za-z0-9/_-]+)", txt)))
    return {"nranks_logged": nranks[-1] if nranks else None, "transports": via[:6],
            "channels": len(set(re.findall(r"Channel (\d+)", txt)))}


def rccl_self_check(report, world, ranks_seen=None):
    """Under backend=nccl the communicator RCCL built must span every rank.  Two witnesses: the all-reduce of ones every
    rank took part in (`ranks_seen`), and RCCL's own init log.  A witness that contradicts `world` fails the run - the caller
    exits non-zero, which `self_launch` / torchrun turn into a failed job.  A log that is missing or holds no init record
    (NCCL_DEBUG preset by the environment, another log format) fails the run only when there is no all-reduce witness
    either: a working 8-GPU job must not die of a parser.  Returns the message or None."""
    if ranks_seen is not None and ranks_seen != world:
        return f"the all-reduce over the communicator saw {ranks_seen} ranks, the job has {world}"
    logged = None if report is None else report.get("nranks_logged")
    if logged is None:
        return None if ranks_seen == world else "RCCL wrote no init record and no all-reduce witness exists: the communicator cannot be verified"
    if logged != world:
        return f"RCCL logged a communicator of {logged} ranks, the job has {world}"
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    from stabletriton_amd import parallel, synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.pipeline import DenoiseLoop
    from stabletriton_amd.scheduler import euler_discrete_tables

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rccl_log = rccl_problem = None
    if args.same_device:                 # rehearsal on one card: RCCL cannot put two ranks on one device
        args.backend = "gloo"
    if world_env > 1 and args.backend == "nccl":
        rccl_log = os.path.join(tempfile.gettempdir(), f"st_rccl_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('RANK', '0')}.log")
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,P2P,NET")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
    rank, world, local = parallel.init_from_env(args.backend)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    if args.fp8 and dtype != torch.bfloat16:
        raise SystemExit("--fp8 runs on a bf16 model")
    n_sched = 50

    spec = model_spec(args.model)
    model, t_fill, t_bcast, n_bcast = build_model(dev, dtype, rank, world, spec)
    ranks_seen = world
    if world > 1:
        ones = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"all_reduce saw {ranks_seen} ranks, expected {world}"
    gm = optimize_model(model, cuda_graph=False, fp8=args.fp8)
    mode = args.mode
    if mode == "auto":
        mode = "loop" if (args.steps % n_sched == 0 and args.warmup % n_sched == 0 and args.img2img is None) else "step"
    if args.img2img is not None and mode == "loop":
        raise SystemExit("--img2img starts mid-schedule: use --mode step (or auto)")
    loop = DenoiseLoop(gm, args.batch, latent_hw(args), dtype, dev, euler_discrete_tables(n_sched), cross_dim=spec.cross_dim,
                       pooled_dim=spec.pooled_dim, mode=mode, n_time_ids=spec.n_time_ids)
    x = synth.denoise_inputs(args.batch, latent_hw(args), 1234 + rank, device=dev, cross_dim=spec.cross_dim, pooled_dim=spec.pooled_dim,
                             n_time_ids=spec.n_time_ids)
    cond = (x["encoder_hidden_states"].to(dtype), x["text_embeds"].to(dtype), x["time_ids"].to(dtype))
    loop.set_conditioning(*cond)
    steps_per_image = n_sched
    if args.img2img is None:
        loop.set_noise(x["latent"])
    else:
        init = synth.normal("img2img.init", tuple(x["latent"].shape), 78 + rank, dev) * 0.8
        steps_per_image = loop.set_image(init, x["latent"], args.img2img)

    with torch.no_grad():
        # per-prompt setup (text K/V projections + the 50-row time table), amortised over a trajectory: timed on its own
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loop.set_conditioning(*cond)
        torch.cuda.synchronize(dev)
        prompt_setup_ms = (time.perf_counter() - t0) * 1e3

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

        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "denoise it/s, SDXL UNet 1024x1024 50-step, bs=1 per GPU" if args.model == "base" and args.img2img is None and latent_hw(args) == 128 else
                      f"denoise it/s, SDXL-{args.model} UNet {args.latent * 8}x{(args.latent_w or args.latent) * 8}" + (" img2img" if args.img2img is not None else "") + f", bs={args.batch} per GPU",
            "value": round(world * args.batch * args.steps / elapsed, 3),
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8" if args.fp8 else args.dtype, "data": "synthetic",
            "config": {"workload": f"SDXL-{args.model} UNet, latent {args.latent}x{args.latent_w or args.latent} ({args.latent * 8}x{(args.latent_w or args.latent) * 8} px), bs={args.batch}/GPU, "
                                   + (f"{n_sched}-step Euler-discrete loop" if args.img2img is None else
                                      f"img2img strength {args.img2img} ({steps_per_image} of {n_sched} Euler-discrete steps per image)")
                                   + f", hipGraph mode={mode}, no CFG"
                                   + (", transformer projections e4m3 x e4m3 (everything else bf16)" if args.fp8 else ""),
                       "parallelism": f"prompt-parallel x{world}", "weights": "synthetic seed 0",
                       "weight_broadcasts": n_bcast},
            "finite": finite, "capture_s": round(t_capture, 2), "weights_s": round(t_fill, 2),
            "prompt_setup_ms": round(prompt_setup_ms, 2), "steps_per_image": steps_per_image,
            # (the prompt setup is paid once per IMAGE: an img2img trajectory runs steps_per_image steps, not the whole schedule)
            "it_per_s_incl_prompt_setup": round(world * args.batch * steps_per_image / (steps_per_image * ms_per_step * 1e-3 + prompt_setup_ms * 1e-3), 3),
        }
        if world > 1:
            result["bcast_s"] = round(t_bcast, 3)
            result["ranks_seen"] = ranks_seen
            result["backend"] = args.backend
            if rccl_log:
                result["rccl"] = rccl_report(rccl_log)
                rccl_problem = rccl_self_check(result["rccl"], world, ranks_seen) if rank == 0 else None
                if result["rccl"] is None or result["rccl"].get("nranks_logged") is None:
                    result["rccl_note"] = "no init record in RCCL's log (NCCL_DEBUG preset?): verified by the all-reduce witness only"
        if rank == 0 and world == 1:
            if not args.no_census:
                fam, census_ms, marker_ms, glue_ms = census(loop)
                n_launch = sum(f["launches"] for f in fam.values())
                boundary_ms = max(ms_per_step - census_ms, 0.0) / n_launch
                # the committed rocprofv3 / counter summaries beside the live numbers: those of the same command (bf16 headline,
                # strict mode, refiner img2img fp8, bs=4); other variants of the run carry none
                # a committed rocprof / counter profile is attached only when batch, model, element type and workload all match
                # the command it was taken from (tools/round_profiles.sh): headline, strict bs=1, config #3, refiner img2img fp8
                base16 = args.model == "base" and not args.fp8 and args.img2img is None and dtype == torch.bfloat16
                tag = None
                if base16 and args.batch == 1:
                    tag = ""
                elif base16 and args.batch == 4:
                    tag = "_b4"                     # (BASELINE config #3)
                elif args.model == "base" and not args.fp8 and args.img2img is None and dtype == torch.float32 and args.batch == 1:
                    tag = "_strict"
                elif args.model == "refiner" and args.fp8 and args.img2img is not None and args.batch == 1 and dtype == torch.bfloat16:
                    tag = "_refiner"
                if args.latent != 128 or args.latent_w not in (None, 128):
                    tag = None                      # (the profiles are of the 1024 x 1024 workload)
                plain = tag is not None
                tag = tag or ""
                roofs = {k: roofline_of(k, v, boundary_ms, committed=bool(plain), tag=tag) for k, v in fam.items()}
                if dtype == torch.float32:
                    for k, r in roofs.items():
                        r["kernel"] = STRICT_KERNEL.get(k, r["kernel"])
                dominant = max(fam, key=lambda k: fam[k]["ms"])
                result["roofline"] = roofs[dominant]
                result["kernels"] = sorted(roofs.values(), key=lambda r: -r["ms_per_step"])
                result["census"] = {"kernel_only_step_ms": round(census_ms, 3), "graph_step_ms": round(ms_per_step, 3),
                                    "launches_per_step": n_launch, "boundary_us_per_launch": round(boundary_ms * 1e3, 3),
                                    "torch_glue_ms": round(glue_ms, 3), "event_record_us": round(marker_ms * 1e3, 2)}
            if not args.no_extras:
                try:
                    result.update(extras(args, model, gm, loop, mode, dtype, dev, n_sched, cond, x))
                except Exception as e:          # noqa: BLE001  (never lose the measured headline to a side measurement)
                    result["extras_error"] = {"extras": f"{type(e).__name__}: {e}"[:300]}
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(model, latent_hw(args), spec)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if rccl_problem:
        sys.stderr.write(f"bench.py: {rccl_problem}\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
