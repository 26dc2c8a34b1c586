import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as F
from diffusynth_amd import _lib as L
import hip_helpers as h
torch.manual_seed(0)
B, Cin, Hh, Ww, cout = 1, 32, 8, 64, 192
x = torch.randn(B, Cin, Hh, Ww); w = torch.randn(cout, Cin, 3, 3) * 0.05; b = torch.randn(cout)
dt = L.DS_BF16
xd = h.to_nhwc(x, dt); xq = h.from_nhwc(xd)
want = F.conv2d(xq, w.bfloat16().float(), b, padding=1)
pc = h.PackedConv(w, b, dt, 4)
y, st = h.run_conv(pc, xd, pad=1, act=L.ACT_NONE, want_stats=True)
got = h.from_nhwc(y)
err = (got - want).abs()
print("per-channel max err:", [round(v, 2) for v in err.amax(dim=(0, 2, 3)).tolist()][:64])
print("per-channel max err 96..:", [round(v, 2) for v in err.amax(dim=(0, 2, 3)).tolist()][96:128])
print("per-w max err:", [round(v, 2) for v in err.amax(dim=(0, 1, 2)).tolist()])
print("per-h max err:", [round(v, 2) for v in err.amax(dim=(0, 1, 3)).tolist()])
