#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) for the deep 3x3 layers of the split-precision tier: a same-box UPPER BOUND on what it could gain
(VERDICT r04 "next" item 5 — decide by measurement, not by argument).  MEASUREMENT TOOL, not product code: the GEMM leg uses
torch.bmm (hipBLASLt), which the product never links.

For one layer (default 768 -> 768 @ 64 x 16, U-Net batch 128: the heaviest launch of the step, 2.68 ms in profiles/r04_conv_layers.txt) it times

  direct   the product kernel: conv3x3_halo3<HP> through the C ABI on hi / lo planes (GroupNorm fold, GELU, statistics), fp32 accuracy
           via x_hi w_hi + x_lo w_hi + x_hi w_lo;
  gemm     the 16 transformed-domain GEMMs  M_p[tiles, Cout] = D_p[tiles, Cin] U_p[Cin, Cout]  (tiles = B H W / 4) as ONE batched bf16 GEMM
           with the three split-precision terms concatenated along K ([D_hi | D_lo | D_hi] x [U_hi ; U_hi ; U_lo], fp32 accumulate
           inside the MFMA, bf16 or fp32 out) — the vendor library's rate for exactly this shape: an optimistic stand-in for a
           hand-written kernel, which would also have to write fp32 and stream 4 x the direct kernel's input bytes;
  xform    streaming kernels moving the BYTES of the two transforms and nothing else: input pass reads B H W Cin fp32-equivalents
           (hi / lo planes) and writes 4 x that as planes (16 positions per 2 x 2 tile); output pass reads 16 fp32 per tile and channel
           and writes 4 — a floor: the real passes also do the transform arithmetic, the split and the epilogue.

Winograd's best case = gemm + xform_in + xform_out.  Gate (VERDICT): >= 20 % under `direct`.  Accuracy is not probed here: F(2x2,3x3)'s transforms
are exact in fp32 up to rounding (+-1 and 1/2 coefficients); the question this tool answers is time."""
import argparse
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffusynth_amd import _lib as L  # noqa: E402
import hip_helpers as h  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def direct_us(B, Cin, Cout, H, W, iters):
    from diffusynth_amd.engine import split3_weight
    w = torch.randn(Cout, Cin, 3, 3) * 0.05
    bias = torch.randn(Cout)
    gam, bet = torch.ones(Cin), torch.zeros(Cin)
    pc = h.PackedConv(split3_weight(w, gam), bias, L.DS_BF16, L.TILE_HALO3_256x96)
    t1, t2 = torch.empty(9 * Cout, device="cuda"), torch.empty(9 * Cout, device="cuda")
    wd, gd, bd = w.cuda().contiguous(), gam.cuda(), bet.cuda()
    L.call("ds_conv_fold_tables", wd.data_ptr(), pc.bias.data_ptr(), gd.data_ptr(), bd.data_ptr(), Cout, Cin, 3, 3, t1.data_ptr(), t2.data_ptr(), L.current_stream())
    xs = (torch.randn(B, H, W, 2 * Cin, device="cuda") * 0.7).bfloat16()
    ab = torch.tensor([[1.0, 0.0]] * B, device="cuda")
    out = torch.empty(B, H, W, 2 * Cout, device="cuda", dtype=torch.bfloat16)
    p = L.ConvParams(src0=xs.data_ptr(), src1=None, C0=2 * Cin, C1=0, H=H, W=W, H1=0, W1=0, off_h1=0, off_w1=0, wpk=pc.w.data_ptr(), Cout=Cout,
                     cout_pad=pc.cout_pad, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, Ho=H, Wo=W, transposed=0, out=out.data_ptr(), out_C=2 * Cout,
                     out_c0=0, out_nchw_f32=0, bias=pc.bias.data_ptr(), gn_ab=ab.data_ptr(), fold_t1=t1.data_ptr(), fold_t2=t2.data_ptr(),
                     ncls=9, act=L.ACT_GELU, res=None, stats_part=None, B=B, dtype=L.DS_BF16, tile=L.TILE_HALO3_256x96, wk_order=1, flags=1 | 2)
    parts = L.load().ds_conv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    stream = L.current_stream()
    return timeit(lambda: L.call("ds_conv_igemm", C.byref(p), stream), iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=768)
    ap.add_argument("--cout", type=int, default=768)
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--w", type=int, default=16)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    B, Cin, Cout, H, W = a.batch, a.cin, a.cout, a.h, a.w
    torch.manual_seed(0)
    tiles = B * (H // 2) * (W // 2)
    t_direct = direct_us(B, Cin, Cout, H, W, a.iters)
    alg = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"layer {Cin}->{Cout} @{H}x{W} B={B}: direct split-precision kernel {t_direct:.1f} us = {alg / t_direct / 1e6:.1f} TF effective fp32, "
          f"{3 * alg / t_direct / 1e6:.1f} TF of bf16 MFMA work")
    # --- 16 GEMMs, three terms along K
    D = (torch.randn(16, tiles, 3 * Cin, device="cuda") * 0.5).bfloat16()
    U = (torch.randn(16, 3 * Cin, Cout, device="cuda") * 0.05).bfloat16()
    Mo = torch.empty(16, tiles, Cout, device="cuda", dtype=torch.bfloat16)
    t_gemm = timeit(lambda: torch.bmm(D, U, out=Mo), a.iters)
    gf = 2.0 * 16 * tiles * 3 * Cin * Cout
    print(f"gemm   16 x [{tiles} x {3 * Cin}] x [{3 * Cin} x {Cout}] bf16 (hipBLASLt via torch.bmm, bf16 out): {t_gemm:.1f} us = {gf / t_gemm / 1e6:.1f} TF "
          f"({gf / 1e9:.0f} GFLOP of MFMA work = 1 / 2.25 of the direct kernel's {3 * alg / 1e9:.0f})")
    del D, U, Mo
    torch.cuda.empty_cache()
    # --- bytes of the transforms as streaming passes
    xin = torch.empty(B * H * W * Cin, device="cuda", dtype=torch.float32)                    # hi / lo planes = 4 B per element
    dt = torch.empty(4, B * H * W * Cin, device="cuda", dtype=torch.float32)                  # 16 positions per 2x2 tile = 4 x, as planes
    t_in = timeit(lambda: dt.copy_(xin.unsqueeze(0).expand(4, -1)), a.iters)
    mt = torch.empty(4, B * H * W * Cout, device="cuda", dtype=torch.float32)                 # 16 fp32 per tile and channel = 4 x the output
    yo = torch.empty(B * H * W * Cout, device="cuda", dtype=torch.float32)
    t_out = timeit(lambda: torch.sum(mt, 0, out=yo), a.iters)
    print(f"xform  input pass (read {xin.numel() * 4 / 1e6:.0f} MB, write {dt.numel() * 4 / 1e6:.0f} MB): {t_in:.1f} us = {(xin.numel() + dt.numel()) * 4 / t_in / 1e6:.2f} TB/s; "
          f"output pass (read {mt.numel() * 4 / 1e6:.0f} MB, write {yo.numel() * 4 / 1e6:.0f} MB): {t_out:.1f} us = {(mt.numel() + yo.numel()) * 4 / t_out / 1e6:.2f} TB/s")
    best = t_gemm + t_in + t_out
    print(f"winograd best case = gemm + xforms = {best:.1f} us = {best / t_direct:.3f} x direct  ({(1 - best / t_direct) * 100:+.1f} % time saved; gate: >= 20 %)")
    print(f"  (gemm alone {t_gemm / t_direct:.3f} x direct; with the GEMM at the direct kernel's own MFMA rate it would be {t_direct / 2.25:.1f} us -> "
          f"{(t_direct / 2.25 + t_in + t_out) / t_direct:.3f} x direct)")


if __name__ == "__main__":
    main()
