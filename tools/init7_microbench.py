#!/usr/bin/env python3
"""7x7 init-convolution micro-benchmark through the C ABI: python tools/init7_microbench.py --h 256 --w 64 --batch 128"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B, H, W = a.batch, a.h, a.w
    torch.manual_seed(0)
    x = torch.randn(B, H, W, 8, device="cuda").bfloat16()
    w = torch.randn(96, 4, 7, 7, device="cuda") * 0.1
    b = torch.randn(96, device="cuda")
    wp = torch.empty(L.load().ds_conv7x7_c4_weight_elems(), dtype=torch.bfloat16, device="cuda")
    out = torch.empty(B, H, W, 96, dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_conv7x7_c4", w.data_ptr(), 96, 4, wp.data_ptr(), st)
    run = lambda: L.call("ds_conv7x7_c4", x.data_ptr(), B, H, W, 8, wp.data_ptr(), b.data_ptr(), out.data_ptr(), st)
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
    mb = out.numel() * 2 / 1e6
    print(f"conv7x7_c4 {H}x{W} B={B}: {us:.1f} us  {mb / us:.2f} TB/s written ({mb:.0f} MB), {2 * B * H * W * 96 * 196 / us / 1e6:.0f} TF")


if __name__ == "__main__":
    main()
