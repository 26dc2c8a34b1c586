#!/usr/bin/env python3
"""Localise a failure of the production U-Net to a kernel family: one forward per engine A/B switch (each in a fresh process).
    DS_LIB=libdiffusynth_hip_bounds.so python tools/debug_switches.py bf16 16 256 64"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
CODE = r'''
import sys, json, torch
sys.path.insert(0, "%s")
import ctypes, diffusynth_amd._lib as LL          # (an older library of a bisect run may lack newer entry points)
_l = ctypes.CDLL(LL.lib_path()); LL._PROTOS = {k: v for k, v in LL._PROTOS.items() if hasattr(_l, k)}
from diffusynth_amd.synth import synth_input, synth_state_dict
from diffusynth_amd.unet import ConditionedUnet, PRODUCTION_CONFIG
keys = json.load(open("%s/tests/golden/state_dict_keys.json"))
net = ConditionedUnet(**PRODUCTION_CONFIG)
net.load_state_dict(synth_state_dict([(k, tuple(s)) for k, s in keys["unet_production"]])); net.to("cuda"); net.set_compute_dtype("%s")
B, H, W = %d, %d, %d
y = net(synth_input("bs_x", (B, 4, H, W)).cuda(), torch.arange(B).cuda() * 37 %% 1000, synth_input("bs_c", (B, 512)).cuda())
bad = (~torch.isfinite(y)).flatten(1).any(1).nonzero().flatten().tolist()
print("finite", bool(torch.isfinite(y).all()), "bad samples", bad[:8], "absmax %%.3g" %% y[torch.isfinite(y)].abs().max().item())
''' % (ROOT, ROOT, dt, B, H, W)
for sw in os.environ.get("DS_DEBUG_SW", "").split(",") if os.environ.get("DS_DEBUG_SW") is not None else ["", "DS_NO_HALO", "DS_NO_SPLITK", "DS_ATTN_V1", "DS_NO_FUSED_ATTN", "DS_NO_COND_ASYNC", "DS_NO_DW_MFMA", "DS_NO_QUAD", "DS_NO_SMALLN", "DS_NO_RESFUSE", "DS_NO_LAZY_GN"]:
    env = dict(os.environ)
    if sw:
        env[sw] = "1"
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    err = [l for l in r.stderr.strip().splitlines() if "amdgpu.ids" not in l]
    print(f"{sw or 'default':18s}", (r.stdout.strip().splitlines() or ["?"])[-1], (err or [""])[-1][:160], flush=True)
