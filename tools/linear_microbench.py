#!/usr/bin/env python3
"""ds_linear micro-benchmark at the five shapes one U-Net step launches (batch = 2 x clips under classifier-free guidance):
python tools/linear_microbench.py --batch 128"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    B = a.batch
    st = L.current_stream()
    # (name, K, O, act_in): time MLP, stacked per-block time biases, condition embedding, stacked label key / value projections
    for name, K, O, act in (("time_mlp.1", 96, 384, 0), ("time_mlp.3", 384, 384, 1), ("time biases (44 blocks)", 384, 13440, 1),
                            ("cond embed", 512, 512, 0), ("label key/value (16 blocks)", 512, 4096, 0)):
        x = torch.randn(B, K, device="cuda")
        w = torch.randn(O, K, device="cuda") * 0.05
        b = torch.randn(O, device="cuda")
        y = torch.empty(B, O, device="cuda")
        run = lambda: L.call("ds_linear", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), B, K, O, act, y.data_ptr(), O, st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        ref = torch.nn.functional.linear(torch.nn.functional.gelu(x) if act else x, w, b)
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        print(f"{name:30s} B={B} K={K} O={O}: {us:7.1f} us   weights {O * K * 4 / 1e6:.1f} MB   rel err vs torch {err:.1e}")


if __name__ == "__main__":
    main()
