#!/usr/bin/env python3
"""ds_conv3x3_c80 at the decoder's shape (B x 256 x 128 x 80): with the GroupNorm + activation applied on load (what the plan runs) and as a plain
convolution + residual (no staging arithmetic) — how much of the kernel is the staging VALU work.   python tools/c80_microbench.py --batch 64"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=128)
    a = ap.parse_args()
    B, H, W, G = a.batch, a.h, a.w, 16
    torch.manual_seed(0)
    x = torch.randn(B, H, W, 80, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    w = (torch.randn(80, 80, 3, 3, device="cuda") * 0.05).contiguous()
    bias, gamma, beta = torch.randn(80, device="cuda"), torch.ones(80, device="cuda"), torch.zeros(80, device="cuda")
    wp = torch.empty(L.load().ds_conv3x3_c80_weight_elems(), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_conv3x3_c80", w.data_ptr(), 80, 80, wp.data_ptr(), st)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats", x.data_ptr(), L.DS_BF16, B, H * W, 80, G, 1e-6, ab.data_ptr(), st)
    slots = L.load().ds_conv3x3_c80_stats_slots(B, H, W)
    ws = torch.empty(B, slots, 80, 2, device="cuda")
    for name, args in (("GroupNorm + swish on load", (ab.data_ptr(), G, gamma.data_ptr(), beta.data_ptr(), L.ACT_SILU, 1, ws.data_ptr())),
                       ("plain + residual", (None, 0, None, None, L.ACT_NONE, 1, ws.data_ptr()))):
        for _ in range(2):
            L.call("ds_conv3x3_c80", x.data_ptr(), B, H, W, wp.data_ptr(), bias.data_ptr(), out.data_ptr(), *args, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.call("ds_conv3x3_c80", x.data_ptr(), B, H, W, wp.data_ptr(), bias.data_ptr(), out.data_ptr(), *args, st)
        e1.record()
        torch.cuda.synchronize()
        print(f"conv3x3_c80 B={B} {H}x{W} {name}: {e0.elapsed_time(e1) * 100:.1f} us")


def convt(B, H, W, cin):
    """ds_convt4x4_c80 (ConvTranspose2d(cin, 80, 4, 2, 1)) with GroupNorm + ReLU on load vs plain."""
    G = 16
    x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, 2 * H, 2 * W, 80, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(cin, 80, 4, 4, device="cuda") * 0.05).contiguous()
    bias, gamma, beta = torch.randn(80, device="cuda"), torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    wp = torch.empty(L.load().ds_convt4x4_c80_weight_elems(cin), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_convt4x4_c80", w.data_ptr(), cin, 80, wp.data_ptr(), st)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats", x.data_ptr(), L.DS_BF16, B, H * W, cin, G, 1e-6, ab.data_ptr(), st)
    ws = torch.empty(B, L.load().ds_convt4x4_c80_stats_slots(B, H, W, cin), 80, 2, device="cuda")
    for name, args in (("GroupNorm + ReLU on load", (ab.data_ptr(), G, gamma.data_ptr(), beta.data_ptr(), ws.data_ptr())), ("plain", (None, 0, None, None, ws.data_ptr()))):
        for _ in range(2):
            L.call("ds_convt4x4_c80", x.data_ptr(), B, H, W, cin, wp.data_ptr(), bias.data_ptr(), out.data_ptr(), *args, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.call("ds_convt4x4_c80", x.data_ptr(), B, H, W, cin, wp.data_ptr(), bias.data_ptr(), out.data_ptr(), *args, st)
        e1.record()
        torch.cuda.synchronize()
        print(f"convt4x4_c80 B={B} {H}x{W} {cin}->80 {name}: {e0.elapsed_time(e1) * 100:.1f} us")


if __name__ == "__main__":
    main()
    convt(64, 256, 128, 80)
    convt(64, 128, 64, 160)
