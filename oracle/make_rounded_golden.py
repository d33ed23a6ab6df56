"""TEST INFRASTRUCTURE - what 16-bit STORAGE alone costs (tests/golden/*_rounded.npz).

Needs no reference and no GPU: the oracle (pinned bit for bit to the reference by tests/test_oracle_golden.py) is run with
every weight and every operator result rounded to bf16 / fp16 and back (`unet_oracle.storage`), fp32 arithmetic inside the
operators, on the inputs of the F1 / F3 fixtures.  Recorded: the deviation of those runs from the REFERENCE's fp32 outputs
(the committed goldens) - max abs and rms - plus the rounded outputs themselves.  tests/test_unet_gpu.py bounds the HIP
path's own bf16 / fp16 deviation by a small factor of these numbers instead of by "1.5 x what we measured".

    python oracle/make_rounded_golden.py f1            # ~1 min
    python oracle/make_rounded_golden.py f3_64         # ~10 min (two 50-step trajectories at latent 64)
    python oracle/make_rounded_golden.py f3_128        # ~45 min (latent 128)
    python oracle/make_rounded_golden.py f3_cfg        # ~40 min (the Diffusers call-site protocol: fp16 pipeline, CFG batch 2, latent 64)
    python oracle/make_rounded_golden.py f1_fp8        # ~1 min (bf16 storage + the fp8 plan's e4m3 operands: what the fp8 mode's format costs on F1)
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stabletriton_amd import synth                     # noqa: E402
from stabletriton_amd.scheduler import euler_discrete_tables  # noqa: E402
from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel  # noqa: E402
from oracle import unet_oracle as orc                  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


def weights():
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE)
    m = m.to_empty(device="cpu").float().eval().requires_grad_(False)
    synth.fill_module_(m, 0)
    return {k: v.detach() for k, v in m.state_dict().items()}


def stats(out, ref):
    d = (out - ref)
    return float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt()), float(ref.abs().max())


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()},
                        meta_torch_version=np.asarray(torch.__version__), meta_generated=np.asarray(time.strftime("%Y-%m-%d")))
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def run_f1(sd):
    g = np.load(os.path.join(OUT, "f1_unet_step_latent64.npz"))
    ref = torch.from_numpy(g["out"])
    x = synth.denoise_inputs(1, 64, 1234)
    rec = {}
    for name, dt in TYPES.items():
        sdr = orc.rounded_state_dict(sd, dt)
        xr = {k: v.to(dt).float() for k, v in x.items()}
        with torch.no_grad(), orc.storage(dt):
            out = orc.unet_forward(sdr, xr["latent"], torch.tensor(999.0), xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"])
        mx, rms, ref_rms, ref_max = stats(out, ref)
        print(f"F1 {name}: max abs {mx:.4f} rms {rms:.5f} (ref rms {ref_rms:.4f}, |ref| max {ref_max:.2f})", flush=True)
        rec[f"{name}_max_abs"], rec[f"{name}_rms"] = mx, rms
        rec[f"{name}_out"] = out[0, :, ::4, ::4].contiguous()
    rec["ref_rms"], rec["ref_max"] = ref_rms, ref_max
    save("f1_unet_step_latent64_rounded", **rec)


def run_f1_fp8(sd):
    """bf16 storage AND the projections of the fp8 plan on e4m3 operands (unet_oracle.fp8_plan): the deviation from the reference's
    fp32 output that the fp8 mode's FORMAT causes on F1, before any kernel is involved."""
    g = np.load(os.path.join(OUT, "f1_unet_step_latent64.npz"))
    ref = torch.from_numpy(g["out"])
    x = synth.denoise_inputs(1, 64, 1234)
    dt = torch.bfloat16
    sdr = orc.rounded_state_dict(sd, dt)
    xr = {k: v.to(dt).float() for k, v in x.items()}
    with torch.no_grad(), orc.storage(dt), orc.fp8_plan():
        out = orc.unet_forward(sdr, xr["latent"], torch.tensor(999.0), xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"])
    mx, rms, ref_rms, ref_max = stats(out, ref)
    print(f"F1 bf16 storage + fp8 plan: max abs {mx:.4f} rms {rms:.5f} = {rms / ref_rms:.3f} of the reference's rms {ref_rms:.4f} (|ref| max {ref_max:.2f})", flush=True)
    save("f1_unet_step_latent64_fp8plan", max_abs=mx, rms=rms, rel_rms=rms / ref_rms, ref_rms=ref_rms, ref_max=ref_max, out=out[0, :, ::4, ::4].contiguous())


def run_f3(sd, hw):
    g = np.load(os.path.join(OUT, f"f3_euler50_latent{hw}.npz"))
    ref = torch.from_numpy(g["final"] if "final" in g else g["latent"])
    tables = euler_discrete_tables(50)
    x = synth.denoise_inputs(1, hw, 1234)
    rec = {}
    for name, dt in TYPES.items():
        sdr = orc.rounded_state_dict(sd, dt)
        xr = {k: v.to(dt).float() for k, v in x.items()}
        t0 = time.time()
        with torch.no_grad(), orc.storage(dt):
            # the loop state stays fp32 (as in DenoiseLoop); the UNet input is the rounded scaled latent
            out = orc.euler_denoise(lambda xi, t: orc.unet_forward(sdr, xi.to(dt).float(), t, xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"]),
                                    x["latent"], tables)
        mx, rms, ref_rms, ref_max = stats(out, ref)
        print(f"F3 latent {hw} {name}: max abs {mx:.4f} rms {rms:.5f} (ref rms {ref_rms:.4f}, |ref| max {ref_max:.2f}) in {time.time() - t0:.0f} s", flush=True)
        rec[f"{name}_max_abs"], rec[f"{name}_rms"] = mx, rms
        rec[f"{name}_final"] = out[0, :, ::4, ::4].contiguous()
    rec["ref_rms"], rec["ref_max"] = ref_rms, ref_max
    save(f"f3_euler50_latent{hw}_rounded", **rec)


def run_f3_cfg(sd):
    """The reference call site's protocol (tests/test_hooks_gpu.py StubPipeline): an fp16 pipeline (fp16 latent state, fp16
    tensors at the UNet boundary, guidance 9 on the difference of two UNet outputs) around a UNet that stores bf16 / fp16."""
    g = np.load(os.path.join(OUT, "f3_cfg50_latent64.npz"))
    ref = torch.from_numpy(g["final"])
    gs = float(g["guidance_scale"])
    tables = euler_discrete_tables(50)
    x = synth.denoise_inputs(2, 64, 1234)
    h = lambda t: t.half().float()
    rec = {}
    for name, dt in TYPES.items():
        sdr = orc.rounded_state_dict(sd, dt)
        xr = {k: h(v).to(dt).float() for k, v in x.items()}
        t0 = time.time()
        lat = h(x["latent"][:1].float() * tables.init_noise_sigma)
        in_scale, dsigma = tables.in_scale(), tables.dsigma()
        with torch.no_grad(), orc.storage(dt):
            for i in range(tables.n_steps):
                xin = h(torch.cat([lat, lat]) * float(in_scale[i])).to(dt).float()
                eps2 = h(orc.unet_forward(sdr, xin, torch.tensor(float(tables.timesteps[i])), xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"]))
                eps = h(eps2[0:1] + gs * h(eps2[1:2] - eps2[0:1]))          # fp16 tensor arithmetic of the pipeline
                lat = h(lat + eps * float(dsigma[i]))
        mx, rms, ref_rms, ref_max = stats(lat, ref)
        print(f"F3-cfg latent 64, fp16 pipeline / {name} UNet storage: max abs {mx:.4f} rms {rms:.5f} (ref rms {ref_rms:.4f}, |ref| max {ref_max:.2f}) in {time.time() - t0:.0f} s", flush=True)
        rec[f"{name}_max_abs"], rec[f"{name}_rms"] = mx, rms
        rec[f"{name}_final"] = lat[0, :, ::4, ::4].contiguous()
    rec["ref_rms"], rec["ref_max"] = ref_rms, ref_max
    save("f3_cfg50_latent64_rounded", **rec)


if __name__ == "__main__":
    what = sys.argv[1:] or ["f1"]
    torch.set_num_threads(int(os.environ.get("ST_ORACLE_THREADS", "6")))
    sd = weights()
    if "f1" in what:
        run_f1(sd)
    if "f1_fp8" in what:
        run_f1_fp8(sd)
    if "f3_64" in what:
        run_f3(sd, 64)
    if "f3_128" in what:
        run_f3(sd, 128)
    if "f3_cfg" in what:
        run_f3_cfg(sd)
