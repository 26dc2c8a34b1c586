#!/usr/bin/env python3
"""Localise a tier error on the small linear_cat U-Net (dims 32/64): run it under one A/B switch at a time."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests")
from conftest import golden_keys, load_golden, rel_err
from diffusynth_amd.synth import synth_state_dict
from diffusynth_amd.unet import ConditionedUnet
g = load_golden("unet_variants")
m = ConditionedUnet(in_dim=4, down_dims=[32, 32, 64], up_dims=[64, 64, 32], attn_type="linear_cat", condition_type="natural_language_prompt", label_emb_dim=64)
m.load_state_dict(synth_state_dict(golden_keys("unet_small_cat"))); m.to("cuda"); m.set_compute_dtype("bf16")
x, t, c = (torch.from_numpy(g[k]).cuda() for k in ("cat_x", "cat_t", "cat_c"))
y = m(x, t, c)
print("shape", tuple(x.shape), "err %%.3e finite %%s" %% (rel_err(y.cpu(), g["cat_y"]), bool(torch.isfinite(y).all())))
''' % (ROOT, ROOT)
for sw in ["", "DS_NO_HALO3", "DS_NO_HALO2", "DS_NO_HALO", "DS_NO_DW_MFMA", "DS_NO_QUAD", "DS_NO_SMALLN", "DS_NO_RESFUSE", "DS_NO_LAZY_GN", "DS_NO_SPLITK", "DS_NO_FUSED_ATTN"]:
    env = dict(os.environ)
    if sw:
        env[sw] = "1"
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{sw or 'default':18s}", (r.stdout.strip().splitlines() or ["?"])[-1], (r.stderr.strip().splitlines() or [""])[-1][:150])
