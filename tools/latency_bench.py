#!/usr/bin/env python3
"""Small-batch latency of the sampling loop (the reference UI's regime: batch 1-8, 10-20 steps, gradio_webUI.py:58,69):
eager plan vs HIP-graph replay.   python tools/latency_bench.py --batch 1 --height 128 --steps 20"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from diffusynth_amd.sampler import DiffSynthSampler  # noqa: E402
from diffusynth_amd.synth import synth_input  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--cfg", type=float, default=1.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    net = build_model(a.dtype, dev)
    cond = synth_input("bench_cond", (512,)).to(dev).unsqueeze(0).repeat(a.batch, 1)
    unc = synth_input("bench_uncond", (512,)).to(dev)
    for graph in (False, True):
        net.use_hip_graph(graph)
        res = []
        for rep in range(3):
            s = DiffSynthSampler(1000, mute=True, device=dev, height=a.height, max_batchsize=a.batch, noise_device="philox")
            s.respace(list(np.linspace(0, 999, a.steps, dtype=np.int32)))
            if a.cfg != 1.0:
                s.activate_classifier_free_guidance(a.cfg, unc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.sample(net, (a.batch, 4, a.height, 64), return_tensor=True, condition=cond, sampler="ddim", seed=1)
            torch.cuda.synchronize()
            res.append(time.perf_counter() - t0)
        dt = min(res[1:])
        print(f"B={a.batch} {a.height}x64 {a.dtype} CFG={a.cfg} {a.steps}-step DDIM, {'HIP graph' if graph else 'eager plan'}: "
              f"{dt * 1e3:.1f} ms per sample() = {dt / a.steps * 1e3:.3f} ms/step, {a.batch * a.steps / dt:.1f} steps/s")


if __name__ == "__main__":
    main()
