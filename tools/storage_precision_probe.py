#!/usr/bin/env python3
"""What would a 16-bit STORAGE format cost?  CPU experiment with the oracle (test infrastructure, not the product path):
every convolution / linear / norm / attention output of the U-Net is rounded to the storage type (and the weights once),
arithmetic stays fp32 — the numerics of a kernel chain that keeps activations in that type between launches and
accumulates in fp32 (bf16 = the shipped throughput tier; fp16 = same MFMA rate, 3 more mantissa bits).
    python tools/storage_precision_probe.py
Prints max|d|/max|ref| of one forward pass and of a 5-step DDIM trajectory against the fp32 oracle."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd.synth import synth_input, synth_state_dict  # noqa: E402
from oracle import unet_ref as U  # noqa: E402
from oracle.sampler_ref import RefSampler  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


def run(dtype):
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        spec = [(k, tuple(s)) for k, s in json.load(f)["unet_production"]]
    sd = synth_state_dict(spec)
    rnd = (lambda t: t) if dtype is None else (lambda t: t.to(dtype).float())
    sdq = {k: (rnd(v) if v.dim() > 1 else v) for k, v in sd.items()}
    orig = (F.conv2d, F.linear, F.group_norm, torch.einsum)
    if dtype is not None:
        F.conv2d = lambda *a, **k: rnd(orig[0](*a, **k))
        F.group_norm = lambda *a, **k: rnd(orig[2](*a, **k))
        torch.einsum = lambda *a, **k: rnd(orig[3](*a, **k))
    try:
        x = synth_input("probe_x", (1, 4, 128, 64))
        t = torch.tensor([500])
        c = synth_input("probe_c", (1, 512))
        y = U.unet_forward(sdq, U.PRODUCTION_CONFIG, x, t, c)
        s = RefSampler(1000, height=32, max_batchsize=2)
        s.respace(list(np.linspace(0, 999, 5, dtype=np.int32)))
        model = lambda xx, tt, cc: U.unet_forward(sdq, U.PRODUCTION_CONFIG, xx, tt, cc)
        traj, _ = s.sample(model, (2, 4, 32, 64), condition=synth_input("probe_c2", (2, 512)), sampler="ddim", seed=1234)
    finally:
        F.conv2d, F.linear, F.group_norm, torch.einsum = orig
    return y, traj


if __name__ == "__main__":
    torch.set_num_threads(8)
    y0, t0 = run(None)
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        y, tr = run(dt)
        print(f"{name} storage: forward rel err {rel(y, y0):.2e}; 5-step DDIM per-step rel err {['%.1e' % rel(a, b) for a, b in zip(tr, t0)]}; "
              f"|activation| max seen {max(float(a.abs().max()) for a in tr):.1f}")
