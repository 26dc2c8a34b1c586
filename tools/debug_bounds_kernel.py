#!/usr/bin/env python3
"""Run the fused-res kernel case under the bounds library and describe where the output is wrong (DS_LIB=libdiffusynth_hip_bounds.so)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffusynth_amd import _lib as L
import test_hip_kernels as T
import conftest
lib = L.load()
seen = {}
orig = conftest.rel_err
def spy(a, b):
    a = torch.as_tensor(a); b = torch.as_tensor(b)
    bad = ~torch.isfinite(a)
    idx = bad.nonzero()
    print("   non-finite:", int(bad.sum()), "of", a.numel())
    if len(idx):
        for d, name in enumerate("bchw"):
            vals = idx[:, d].unique().tolist()
            print("    ", name, vals[:40], "..." if len(vals) > 40 else "")
    d = (a.double() - b.double()).abs()
    d[bad] = 0
    print("   max err where finite: %.3e" % (d.max() / b.abs().max()).item())
    return orig(a, b)
T.rel_err = spy
for shape, cx in (((2, 192, 16, 32), (96, 0)), ((2, 96, 9, 27), (96, 96))):
    try:
        T.test_conv3x3_halo2_with_fused_res_conv(shape, cx, L.TILE_HALO3_256x96)
        print(shape, cx, "ok")
    except AssertionError as e:
        print(shape, cx, "FAILED")
