#!/usr/bin/env python3
"""Fused attention micro-benchmark through the C ABI: python tools/attn_microbench.py --c 96 --n 16384 --batch 16"""
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
    ap.add_argument("--c", type=int, default=96)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nseg", type=int, default=0)
    ap.add_argument("--gen", type=int, default=0, help="1 / 2: force the first / second kernel generation (default: second unless DS_ATTN_V1 / DS_ATTN_CTX1 is set)")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B, N, Cc = a.batch, a.n, a.c
    gen = a.gen if a.gen else (1 if (os.environ.get("DS_ATTN_V1") or os.environ.get("DS_ATTN_CTX1")) else 2)      # (forced: the A/B is about the kernels)
    nseg = a.nseg or (max(1, min((1024 if Cc == 384 else 2048) // B, 64, (N + 31) // 32)) if gen == 2 and (Cc in (96, 192) or N >= 1024) else max(1, min(N // 128, 32)))
    torch.manual_seed(0)
    x = torch.randn(B, N, Cc, device="cuda").bfloat16()
    wq16 = (torch.randn(384 * Cc, device="cuda") * Cc ** -0.5).bfloat16()
    wo16 = (torch.randn(Cc * 128, device="cuda") * 128 ** -0.5).bfloat16()
    t1, t2 = torch.randn(384, device="cuda") * 0.1, torch.randn(384, device="cuda") * 0.1
    ab = torch.tensor([[1.0, 0.0]] * B, device="cuda")
    lq = torch.randn(B, 128, device="cuda")
    part = torch.empty(L.load().ds_linattn_part_floats(B, 4, nseg), device="cuda")
    ctx = torch.empty(B * 4 * 1024, device="cuda")
    y = torch.empty(B, N, Cc, dtype=torch.bfloat16, device="cuda")
    bo = torch.randn(Cc, device="cuda")
    p = L.AttnFusedParams(x=x.data_ptr(), B=B, N=N, C=Cc, nseg=nseg, wqkv=wq16.data_ptr(), t1=t1.data_ptr(), t2=t2.data_ptr(),
                          gn_ab=ab.data_ptr(), label_q=lq.data_ptr(), lq_stride=128, scale=32 ** -0.5, part=part.data_ptr(),
                          ctx=ctx.data_ptr(), wout_perm=wo16.data_ptr(), bias_out=bo.data_ptr(), y=y.data_ptr(), stats_part=None)
    mf = torch.empty(B * Cc * 128, dtype=torch.bfloat16, device="cuda")
    p.mfold = mf.data_ptr() if Cc in (96, 192) else None
    p.gen = gen
    parts = L.load().ds_attn_fused_stats_parts(C.byref(p))
    sp = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = sp.data_ptr()
    st = L.current_stream()
    res = {}
    for name in ("ds_attn_fused_context", "ds_attn_fused_output"):
        for _ in range(3):
            L.call(name, C.byref(p), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            L.call(name, C.byref(p), st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / a.iters
    mb = x.numel() * 2 / 1e6
    print(f"attn C={Cc} N={N} B={B} nseg={nseg} parts={parts}: context(+combine) {res['ds_attn_fused_context']:.1f} us, "
          f"output {res['ds_attn_fused_output']:.1f} us  (x = {mb:.1f} MB)")


if __name__ == "__main__":
    main()
