#!/usr/bin/env python3
"""ds_conv1x1_x3 micro-benchmark at the split-precision tier's 1x1 shapes (fp32 in / out; algorithmic bytes = input once + output + residual):
python tools/x3_1x1_microbench.py --batch 128"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd import _lib as L  # noqa: E402
from diffusynth_amd.engine import pack_x3_1x1  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    B = a.batch
    st = L.current_stream()
    # (name, H, W, C0, C1, Cout, residual)
    cases = [("to_qkv 96 @256x64", 256, 64, 96, 0, 384, 0), ("to_out 128->96 +res @256x64", 256, 64, 128, 0, 96, 1),
             ("res_conv 96+96->96 @256x64", 256, 64, 96, 96, 96, 0), ("to_qkv 192 @128x32", 128, 32, 192, 0, 384, 0),
             ("to_out 128->192 +res @128x32", 128, 32, 128, 0, 192, 1), ("res_conv 192+192->192 @128x32", 128, 32, 192, 192, 192, 0),
             ("to_qkv 384 @64x16", 64, 16, 384, 0, 384, 0), ("to_qkv 768 @32x8", 32, 8, 768, 0, 384, 0)]
    for name, H, W, C0, C1, Cout, res in cases:
        x0 = torch.randn(B, H, W, C0, device="cuda")
        x1 = torch.randn(B, H, W, C1, device="cuda") if C1 else None
        w = torch.randn(Cout, C0 + C1, 1, 1, device="cuda") * (C0 + C1) ** -0.5
        bias = torch.randn(Cout, device="cuda")
        wpk, cout_pad = pack_x3_1x1(w, None)
        out = torch.empty(B, H, W, Cout, device="cuda")
        r = torch.randn(B, H, W, Cout, device="cuda") if res else None
        p = L.ConvParams(src0=x0.data_ptr(), src1=L.ptr(x1), C0=C0, C1=C1, H=H, W=W, H1=(H if C1 else 0), W1=(W if C1 else 0), off_h1=0, off_w1=0,
                         wpk=wpk.data_ptr(), Cout=Cout, cout_pad=cout_pad, KH=1, KW=1, stride=1, pad_h=0, pad_w=0, Ho=H, Wo=W, transposed=0,
                         out=out.data_ptr(), out_C=Cout, out_c0=0, out_nchw_f32=0, bias=bias.data_ptr(), gn_ab=None, fold_t1=None, fold_t2=None,
                         ncls=1, act=L.ACT_NONE, res=L.ptr(r), stats_part=None, B=B, dtype=L.DS_BF16, tile=0)
        p.flags = 8 | 4
        run = lambda: L.call("ds_conv1x1_x3", C.byref(p), st)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        gb = B * H * W * (C0 + C1 + Cout * (2 if res else 1)) * 4 / 1e9
        print(f"{name:32s} B={B}: {us:8.1f} us   {gb:5.2f} GB   {gb / us * 1e3:5.2f} TB/s   {2 * 3 * B * H * W * (C0 + C1) * Cout / us / 1e6:6.0f} TF (bf16 MFMA work)")


if __name__ == "__main__":
    main()
