#!/usr/bin/env python3
"""Depthwise 7x7 micro-benchmark through the C ABI: python tools/dw_microbench.py --c 192 --h 256 --w 64 --batch 16"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c", type=int, default=192)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp32split"], help="fp32split: fp32 in, hi / lo bf16 planes out (the bf16x3 tier)")
    ap.add_argument("--strip", type=int, default=0, help="ds_dwconv_params.strip: 0 library's choice, 1 strip kernel, 2 tile kernel")
    a = ap.parse_args()
    B, H, W, Cc = a.batch, a.h, a.w, a.c
    torch.manual_seed(0)
    f32 = a.dtype != "bf16"
    x = torch.randn(B, H, W, Cc, device="cuda")
    if not f32:
        x = x.bfloat16()
    w = torch.randn(Cc, 49, device="cuda") * 0.1
    wt = torch.empty(49 * Cc, device="cuda")
    wexp = torch.empty(Cc * 6 * 64 * 8, dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_dw_weight", w.data_ptr(), Cc, wt.data_ptr(), st)
    L.call("ds_pack_dw_weight_mfma", w.data_ptr(), Cc, wexp.data_ptr(), st)
    bias = torch.randn(Cc, device="cuda")
    tb = torch.randn(B, Cc, device="cuda")
    out = torch.empty_like(x)
    p = L.DwconvParams(src0=x.data_ptr(), src1=None, C0=Cc, C1=0, H=H, W=W, H1=0, W1=0, off_h1=0, off_w1=0, wt=wt.data_ptr(),
                       wexp=wexp.data_ptr(), bias=bias.data_ptr(), tbias=tb.data_ptr(), tb_stride=Cc, out=out.data_ptr(),
                       stats_part=None, B=B, dtype=L.DS_F32 if f32 else L.DS_BF16)
    if a.dtype == "fp32split":
        p.out_split = 1
    p.strip = a.strip
    parts = L.load().ds_dwconv_stats_parts(C.byref(p))
    sp = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = sp.data_ptr()
    for _ in range(3):
        L.call("ds_dwconv7", C.byref(p), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        L.call("ds_dwconv7", C.byref(p), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    lib = L.load()
    mb = 2 * x.numel() * x.element_size() / 1e6
    print(f"dwconv7 {a.dtype} strip={a.strip} C={Cc} {H}x{W} B={B}: {us:.1f} us  {mb / us:.2f} TB/s (in+out {mb:.0f} MB)")


if __name__ == "__main__":
    main()
