#!/usr/bin/env python3
"""Probe: one U-Net forward at batch 128 against two concurrent forwards at batch 64 on two streams (the two halves of a classifier-free-guidance
step are independent; the question is whether MFMA-bound and HBM-bound kernels of the two halves overlap).  python tools/dual_stream_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from diffusynth_amd.synth import synth_input  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tier = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    n1, n2 = build_model(tier, dev), build_model(tier, dev)
    print("tier", tier)
    B, H, W = 64, 256, 64
    x = torch.randn(2 * B, 4, H, W, device=dev)
    t = torch.full((2 * B,), 500, device=dev, dtype=torch.long)
    cond = synth_input("bench_cond", (512,)).to(dev).unsqueeze(0).repeat(2 * B, 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def single():
        return n1(x, t, cond)

    def dual():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = n1(x[:B], t[:B], cond[:B])
        with torch.cuda.stream(s2):
            b = n2(x[B:], t[B:], cond[B:])
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        return a, b

    def seq():
        return n1(x[:B], t[:B], cond[:B]), n2(x[B:], t[B:], cond[B:])

    for name, fn in (("one forward, batch 128", single), ("two forwards, batch 64, one stream", seq), ("two forwards, batch 64, two streams", dual)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:40s}: {dt * 1e3:7.2f} ms per 128 samples")
    a, b = dual()
    r = single()
    torch.cuda.synchronize()
    print("max |dual - single| / max|single| =", ((torch.cat([a, b]) - r).abs().max() / r.abs().max()).item())


if __name__ == "__main__":
    main()
