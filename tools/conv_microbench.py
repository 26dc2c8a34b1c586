#!/usr/bin/env python3
"""Single-layer convolution micro-benchmark through the C ABI (for rocprofv3 / A-B work).
    python tools/conv_microbench.py --cin 96 --cout 192 --h 256 --w 64 --batch 16 --k 3 --tile 4 --iters 20
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffusynth_amd import _lib as L  # noqa: E402
import hip_helpers as h  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=96)
    ap.add_argument("--cout", type=int, default=192)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--tile", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fold", type=int, default=1)
    ap.add_argument("--act", type=int, default=1, help="1 = GELU epilogue, 0 = none")
    ap.add_argument("--stats", type=int, default=1)
    ap.add_argument("--res", type=int, default=0, help="1: add a residual tensor in the epilogue (conv2 of a ConvNeXt block)")
    ap.add_argument("--x3", default="", choices=["", "split", "f32", "f32res"],
                    help="split-precision launch (conv3x3_halo3<HP>): hi / lo planes in; planes out with GELU (conv1), fp32 out, fp32 out + fp32 residual (conv2)")
    ap.add_argument("--stamp", type=int, default=0, help="1: allocate the debug buffer of a -DDS_STAMP=1 build and print per-wave K-loop timing")
    a = ap.parse_args()
    dt = L.DS_BF16 if a.dtype == "bf16" else L.DS_F32
    torch.manual_seed(0)
    w = torch.randn(a.cout, a.cin, a.k, a.k) * 0.05
    b = torch.randn(a.cout)
    g = torch.ones(a.cin) if a.fold else None
    be = torch.zeros(a.cin) if a.fold else None
    pc = h.PackedConv(w, b, dt, a.tile, gamma=g, beta=be)
    x = torch.randn(a.batch, a.h, a.w, a.cin, device="cuda").to(h.TDT[dt])
    ab = torch.tensor([[1.0, 0.0]] * a.batch, device="cuda") if a.fold else None
    pad = a.k // 2
    for _ in range(3):
        h.run_conv(pc, x, pad=pad, gn_ab=ab, act=L.ACT_GELU if a.act else L.ACT_NONE, want_stats=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    # build params once, launch back to back
    B, H, W, C0 = x.shape
    out = torch.empty(B, H, W, a.cout, device="cuda").to(h.TDT[dt])
    p = L.ConvParams(src0=x.data_ptr(), src1=None, C0=C0, C1=0, H=H, W=W, H1=0, W1=0, off_h1=0, off_w1=0, wpk=pc.w.data_ptr(),
                     Cout=pc.Cout, cout_pad=pc.cout_pad, KH=a.k, KW=a.k, stride=1, pad_h=pad, pad_w=pad, Ho=H, Wo=W, transposed=0,
                     out=out.data_ptr(), out_C=pc.Cout, out_c0=0, out_nchw_f32=0, bias=L.ptr(pc.bias), gn_ab=L.ptr(ab),
                     fold_t1=L.ptr(pc.t1) if a.fold else None, fold_t2=L.ptr(pc.t2) if a.fold else None,
                     ncls=pc.ncls if a.fold else 1, act=L.ACT_GELU if a.act else L.ACT_NONE, res=None, stats_part=None, B=B, dtype=dt, tile=a.tile,
                     wk_order=pc.k_order)
    resid = torch.randn(B, H, W, a.cout, device="cuda").to(h.TDT[dt]) if a.res else None
    if resid is not None:
        p.res = resid.data_ptr()
    if a.x3:
        # the split-precision instantiation: input = hi / lo bf16 planes of a fp32 tensor (2 Cin channels), weights per chunk [W_hi | W_lo | W_hi]
        from diffusynth_amd.engine import split3_weight
        assert a.tile == 11 and a.k == 3 and a.dtype == "bf16"
        pc3 = h.PackedConv(split3_weight(w, g), b, L.DS_BF16, a.tile)
        xs = (torch.randn(B, H, W, 2 * a.cin, device="cuda") * 0.7).bfloat16()
        planes = a.x3 == "split"
        out = torch.empty(B, H, W, 2 * a.cout, device="cuda", dtype=torch.bfloat16) if planes else torch.empty(B, H, W, a.cout, device="cuda")
        resid = torch.randn(B, H, W, a.cout, device="cuda") if a.x3 == "f32res" else None
        p.src0, p.C0, p.wpk, p.out, p.out_C = xs.data_ptr(), 2 * a.cin, pc3.w.data_ptr(), out.data_ptr(), (2 * a.cout if planes else a.cout)
        p.flags, p.act, p.res = 1 | (2 if planes else 4), (L.ACT_GELU if planes else L.ACT_NONE), L.ptr(resid)
        x = xs
    parts = L.load().ds_conv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr() if a.stats else None
    dbg = None
    if a.stamp:
        dbg = torch.zeros(B * 64 * 64 * 32, dtype=torch.int64, device="cuda")
        p.slab = dbg.data_ptr()
    s = L.current_stream()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        L.call("ds_conv_igemm", C.byref(p), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    flops = 2.0 * B * H * W * a.cout * a.cin * a.k * a.k
    byts = (x.numel() + out.numel()) * x.element_size()
    if dbg is not None:
        d = dbg.view(-1, 8)
        d = d[d[:, 3] > 0].double()
        tot, lg, bar, n = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
        print(f"stamps: {d.shape[0]} waves, steps {n[0]:.0f}: loop {tot.mean():.0f} cyc ({(tot / n).mean():.0f}/step), "
              f"lgkm wait {(lg / n).mean():.0f}/step, barrier wait {(bar / n).mean():.0f}/step "
              f"(min {(bar / n).min():.0f} max {(bar / n).max():.0f})")
        pro, loop, k0, tot_rt = d[:, 4] / 100, d[:, 5] / 100, d[:, 6] / 100, d[:, 7] / 100      # 100 MHz -> us
        e1, e2 = d[:, 1] / 100, d[:, 2] / 100
        if a.tile == 11 and a.stamp == 2:
          print(f"        epilogue split: body issued {e1.mean():.2f} us, store drain {e2.mean():.2f} us, sync + statistics + exit {(tot_rt - pro - loop - e1 - e2).mean():.2f} us")
        elif a.tile in (10, 11):
          print(f"        prologue split: setup + small loads issued {e1.mean():.2f} us, statistics + shift table done {e2.mean():.2f} us, loop starts {pro.mean():.2f} us")
        else:
          print(f"        epilogue split: shift table + sync {(e1 - pro - loop).mean():.1f} us, body {(e2 - e1).mean():.1f} us, "
              f"stats + tail {(tot_rt - e2).mean():.1f} us")
        qs = torch.quantile(loop, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=loop.dtype, device=loop.device)).tolist()
        print("        K loop wall per wave, 5 / 25 / 50 / 75 / 95 %: " + " / ".join(f"{v:.1f}" for v in qs) + " us")
        print(f"        wall per wave: prologue {pro.mean():.1f} us, K loop {loop.mean():.1f} us, epilogue {(tot_rt - pro - loop).mean():.1f} us, "
              f"total {tot_rt.mean():.1f} us; kernel span (first start -> last end) {(k0 + tot_rt).max() - k0.min():.1f} us; "
              f"start spread {k0.max() - k0.min():.1f} us; in-loop clock {(tot / (loop * 1e-6)).mean() / 1e9:.2f} GHz")
    print(f"tile {a.tile} {a.k}x{a.k} {a.cin}->{a.cout} @{H}x{W} B={B} {a.dtype}{' x3:' + a.x3 if a.x3 else ''}: {us:.1f} us  {flops / us / 1e6:.1f} TF  "
          f"{byts / us / 1e6:.2f} TB/s(alg in+out)")


if __name__ == "__main__":
    main()
