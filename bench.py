#!/usr/bin/env python3
"""bench.py — denoising-steps/sec of the diffusynth sampling hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch: one DiffSynthSampler.p_sample (U-Net
evaluation(s) + fused DDPM/DDIM update) for the B samples a GPU owns.  The default workload is
BASELINE.json configs[1]: production ConditionedUnet (random init), bf16, batch 16 per GPU,
(4,256,64) latents, text condition given, CFG=1, DDPM steps of a 50-step respaced schedule,
device-side Philox noise (inputs resident in HBM).  value = B_total * K / max-over-ranks seconds.
One JSON line is printed by rank 0; it also carries
  roofline      — the dominant kernel (MFMA implicit-GEMM conv), algorithmic FLOPs / HIP-event time,
  cpu_baseline  — the CPU oracle (a port of the reference's CPU path) timed on this host (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE_NAMES = {0: "conv_igemm<128x192>", 1: "conv_igemm<256x96>", 2: "conv_igemm<128x32>", 3: "conv_igemm<64x192>",
              4: "conv3x3_halo<256x192>", 5: "conv3x3_halo<256x96>", 6: "conv3x3_halo<128x192>", 7: "conv3x3_halo<128x96>",
              8: "conv3x3_halo<256x192,4w>", 9: "conv3x3_halo<256x96,4w>"}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md chip table
WORKLOADS = {
    # name: (batch per GPU, cfg scale, sampler, K of the respaced schedule, conditioned)
    "config2": (16, 1.0, "ddpm", 50, True),        # BASELINE configs[1]
    "config3": (64, 6.0, "ddpm", 50, True),        # BASELINE configs[2]: CFG doubles the U-Net batch
    "config4": (64, 6.0, "ddim", 100, True),       # BASELINE configs[3]: per-GPU share of batch 512 on 8 GPUs
    "config1": (1, 1.0, "ddpm", 50, False),        # BASELINE configs[0] on the GPU
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override batch per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=40, help="CPU oracle steps timed for cpu_baseline")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events in the timed region")
    return ap.parse_args()


def build_model(dtype, device):
    from diffusynth_amd.unet import PRODUCTION_CONFIG, ConditionedUnet
    torch.manual_seed(0)                      # default torch init, same on every rank
    net = ConditionedUnet(**PRODUCTION_CONFIG)
    net.to(device)
    net.set_compute_dtype(dtype)
    return net


def cpu_baseline(H, W, nsteps):
    """The oracle (CPU port of the reference path) on this host: config-1 style, B=1, DDPM, null cond."""
    from diffusynth_amd.unet import PRODUCTION_CONFIG, ConditionedUnet
    from oracle.sampler_ref import RefSampler
    from oracle.unet_ref import RefUnet
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in ConditionedUnet(**PRODUCTION_CONFIG).state_dict().items()}
    model = RefUnet(sd)
    torch.set_num_threads(min(16, os.cpu_count() or 16))     # 16 threads measured fastest on the GPU box's host (8/16/32/64 probed)
    cores = torch.get_num_threads()
    s = RefSampler(1000, height=H, max_batchsize=1)
    s.respace(list(np.linspace(0, 999, 50, dtype=np.int32)))
    torch.manual_seed(1234)
    x, _ = s.noise(1, W)
    x = s.step(model, x, torch.full((1,), 49, dtype=torch.long), None, 1.0)          # warm-up step
    t0 = time.perf_counter()
    for i in range(nsteps):
        x = s.step(model, x, torch.full((1,), 48 - i, dtype=torch.long), None, 1.0)
    dt = time.perf_counter() - t0
    return {"value": nsteps / dt, "unit": "denoising-steps/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} DDPM steps of the 50-step schedule, B=1, (4,{H},{W}) latent, fp32, null condition "
                      f"(oracle/: CPU restatement of the reference path), {dt:.1f} s"}


def main():
    a = parse()
    from diffusynth_amd import dist as D
    from diffusynth_amd.sampler import DiffSynthSampler
    from diffusynth_amd.synth import synth_input
    rank, world, device = D.init()
    assert device.type == "cuda", "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1"
    B, cfg, sampler_name, K, conditioned = WORKLOADS[a.workload]
    if a.batch is not None:
        B = a.batch
    H, W = a.height, a.width
    net = build_model(a.dtype, device)

    # text embeddings live on rank 0 and reach the other ranks by ONE RCCL broadcast
    cond = uncond = None
    if conditioned:
        c0 = synth_input("bench_cond", (512,)) if rank == 0 else None
        u0 = synth_input("bench_uncond", (512,)) if (rank == 0 and cfg != 1.0) else None
        cond, uncond = D.broadcast_conditions(c0, u0, device)
        cond = cond.unsqueeze(0).repeat(B, 1)
    s = DiffSynthSampler(1000, mute=True, device=device, height=H, max_batchsize=B, noise_device="philox")
    s.respace(list(np.linspace(0, 999, K, dtype=np.int32)))
    if cfg != 1.0:
        s.activate_classifier_free_guidance(cfg, uncond)
    s._seed(1234 + rank)
    eta = 0.0 if sampler_name == "ddim" else 1.0
    x, _ = s.get_deterministic_noise_tensor(B, W)
    total = a.warmup + a.steps
    assert total <= K, f"warmup+steps={total} exceeds the {K}-step schedule"
    ts = [torch.full((B,), K - 1 - i, device=device, dtype=torch.long) for i in range(total)]
    coefs = [s._step_coefficients(t.cpu(), eta).to(device) for t in ts]

    def step(i, x):
        return s.ddim_sample(net, x, ts[i], condition=cond, ddim_eta=eta, _coef=coefs[i])

    for i in range(a.warmup):
        x = step(i, x)
    plan = next(iter(net._engine.plans.values()))
    use_events = not a.no_kernel_events
    # per-launch HIP events cost ~6 % of a step when every convolution of every step carries a pair (the markers
    # serialise the queue), so they are recorded on every 4th timed step only: still live, inside the timed region
    prof = [] if use_events else None
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, total):
        plan.prof = prof if (use_events and (i - a.warmup) % 4 == 0) else None
        x = step(i, x)
    torch.cuda.synchronize()
    plan.prof = prof
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, device)
    assert torch.isfinite(x).all(), "non-finite latents"

    # per-launch HIP events (recorded on the launch stream inside the timed region) -> dominant kernel roofline
    roof = None
    if use_events:
        per = {}
        for k, e0, e1 in plan.prof:
            tile, flops, desc = plan.conv_meta[k]
            d = per.setdefault(tile, [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += flops
            d[2] += 1
        if os.environ.get("DS_BENCH_DUMP") and rank == 0:
            agg = {}
            for k, e0, e1 in plan.prof:
                tile, flops, desc = plan.conv_meta[k]
                d = agg.setdefault((k, tile, desc), [0.0, flops, 0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
            for (k, tile, desc), (sec, flops, n) in sorted(agg.items()):
                print(f"  op{k:4d} {TILE_NAMES[tile]:24s} {desc:28s} {sec / n * 1e6:8.1f} us  {flops / (sec / n) / 1e12:7.1f} TF", file=sys.stderr)
        plan.prof = None
        if per:
            tile, (sec, flops, n) = max(per.items(), key=lambda kv: kv[1][0])
            conv_s = sum(v[0] for v in per.values())
            ach = flops / sec / 1e12
            # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
            # MI355X_MICROARCH.md §HBM); collected offline with the same command, so null if the file is absent
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
                    pk = {"conv3x3_halo<256x192>": "conv3x3_halo_kernel<256,192,4,2,2,6>", "conv3x3_halo<256x96>": "conv3x3_halo_kernel<256,96,8,1,2,6>",
                          "conv3x3_halo<256x96,4w>": "conv3x3_halo_kernel<256,96,4,1,2,5>",
                          "conv_igemm<128x192>": "conv_igemm_kernel<bf16,128,192,2,2>", "conv_igemm<64x192>": "conv_igemm_kernel<bf16,64,192,2,2>",
                          "conv_igemm<256x96>": "conv_igemm_kernel<bf16,256,96,4,1>"}.get(TILE_NAMES[tile])
                    kern = {k.replace(" ", ""): v for k, v in json.load(f)["kernels"].items()}
                    if a.dtype == "bf16" and a.workload == "config2" and a.batch is None:
                        traffic = kern[pk]["hbm_bytes"]
            except Exception:
                traffic = None
            roof = {"bound": "mfma", "kernel": TILE_NAMES[tile], "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype],
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[a.dtype], 4), "traffic": traffic,
                    "launches": n, "avg_launch_us": round(sec / n * 1e6, 2), "alg_gflop_per_launch": round(flops / n / 1e9, 3),
                    "all_conv_tflops": round(sum(v[1] for v in per.values()) / conv_s / 1e12, 2),
                    "conv_share_of_step_time": round(conv_s / (elapsed * len(range(0, a.steps, 4)) / a.steps), 3),
                    "event_sampling": "every 4th timed step"}

    if rank != 0:
        return
    evals = 2 if cfg != 1.0 else 1
    value = B * world * a.steps / elapsed
    # whole-step algorithmic roofline: 273 GFLOP and 513 MB (bf16) / 1026 MB (fp32) per sample-eval at 256x64
    scale = (H * W) / (256 * 64)
    flop_s = 273e9 * scale * evals
    byte_s = ((513e6 if a.dtype == "bf16" else 1026e6) * scale + (214e6 if a.dtype == "bf16" else 428e6) / (B * evals)) * evals
    t_mfma, t_hbm = flop_s / (PEAK_TFLOPS[a.dtype] * 1e12), byte_s / 8e12
    out = {
        "metric": "denoising-steps/sec (batch x T) on 256x64 latents", "value": round(value, 2), "unit": "denoising-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{ {'config1': 0, 'config2': 1, 'config3': 2, 'config4': 3}[a.workload] }]: "
                               f"production ConditionedUnet (random init, torch.manual_seed(0)), {a.dtype}, batch {B}/GPU, "
                               f"latent (4,{H},{W}), {'text condition' if conditioned else 'null condition'}, CFG={cfg}, "
                               f"{sampler_name} steps of a {K}-step respaced schedule, Philox noise on device",
                   "global_batch": B * world, "unet_evals_per_step": evals, "parallelism": f"batch-shard x{world}, weights replicated"},
        "step_roofline": {"t_mfma_us_per_sample_step": round(t_mfma * 1e6, 1), "t_hbm_us_per_sample_step": round(t_hbm * 1e6, 1),
                          "frac_of_hbm_roofline": round(t_hbm / (elapsed / a.steps / B), 4),
                          "frac_of_mfma_roofline": round(t_mfma / (elapsed / a.steps / B), 4)},
        "roofline": roof,
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H, W, a.cpu_steps)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
