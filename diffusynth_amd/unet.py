"""ConditionedUnet — drop-in for the reference noise predictor (model/diffusion.py:21-258).

Same constructor arguments, same ``forward(x, time, condition=None)`` contract and the same
state-dict names/shapes (checkpoints of the reference load with ``load_state_dict``), but the
forward pass is a plan of hand-written HIP kernels (libdiffusynth_hip.so) over channels-last
activations:

    ConvNeXt block (components:107-139)  = dwconv7(+bias+time bias, GN partials)
                                           -> 3x3 implicit GEMM [GN folded, GELU, GN partials]
                                           -> (1x1 res_conv) -> 3x3 implicit GEMM [GN folded, +residual]
    attention block (components:22-29,142-152,252-293)
                                         = 1x1 qkv GEMM [PreNorm folded] -> linear-attention context
                                           -> output pass -> 1x1 to_out GEMM [GN partials]
                                           -> GN apply + residual
    skip concat (components:236-249)     = never materialised (kernels read two sources)

The module holds parameters only; there is no PyTorch fallback: on a machine without the HIP
library or without a GPU ``forward`` raises.
"""
import ctypes as C
import math
import os

import torch
from torch import nn

from . import _lib as L


# =============================================================================== parameter tree
class _Holder(nn.Module):
    """Parameter container: children are attached by name; computation lives in the HIP plan."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            setattr(self, k, v)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder; call ConditionedUnet.forward")


def _seq(*mods):
    return nn.Sequential(*mods)


def _convnext_params(dim, dim_out, time_dim, mult):
    h = _Holder()
    h.mlp = _seq(nn.GELU(), nn.Linear(time_dim, dim)) if time_dim is not None else None
    h.ds_conv = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
    h.net = _seq(nn.GroupNorm(1, dim), nn.Conv2d(dim, dim_out * mult, 3, padding=1), nn.GELU(),
                 nn.GroupNorm(1, dim_out * mult), nn.Conv2d(dim_out * mult, dim_out, 3, padding=1))
    h.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
    return h


def _resnet_params(dim, dim_out, time_dim, groups):
    h = _Holder()
    h.mlp = _seq(nn.SiLU(), nn.Linear(time_dim, dim_out)) if time_dim is not None else None
    h.block1 = _Holder(proj=nn.Conv2d(dim, dim_out, 3, padding=1), norm=nn.GroupNorm(groups, dim_out), act=nn.SiLU())
    h.block2 = _Holder(proj=nn.Conv2d(dim_out, dim_out, 3, padding=1), norm=nn.GroupNorm(groups, dim_out), act=nn.SiLU())
    h.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
    return h


def _attn_params(dim, kind, label_emb_dim, heads=4, dim_head=32):
    hid = heads * dim_head
    a = _Holder()
    a.to_qkv = nn.Conv2d(dim, hid * 3, 1, bias=False)
    a.to_out = _seq(nn.Conv2d(hid, dim, 1), nn.GroupNorm(1, dim))
    a.label_key = nn.Linear(label_emb_dim, hid)
    if kind == "linear_add":
        a.label_query = nn.Linear(label_emb_dim, hid)
    else:
        a.label_value = nn.Linear(label_emb_dim, hid)
    return _Holder(fn=_Holder(fn=a, norm=nn.GroupNorm(1, dim)))


class ConditionedUnet(nn.Module):
    """See module docstring.  Signature = model/diffusion.py:22-38."""

    def __init__(self, in_dim, out_dim=None, down_dims=None, up_dims=None, mid_depth=3, with_time_emb=True,
                 time_dim=None, resnet_block_groups=8, use_convnext=True, convnext_mult=2, attn_type="linear_cat",
                 n_label_class=11, condition_type="instrument_family", label_emb_dim=128):
        super().__init__()
        if condition_type == "instrument_family":
            emb = nn.Embedding(int(n_label_class + 1), int(label_emb_dim))
        elif condition_type == "natural_language_prompt":
            emb = nn.Linear(int(label_emb_dim), int(label_emb_dim), bias=True)
        else:
            raise NotImplementedError()
        self.label_embedding = _Holder(embedding=emb)
        up_dims = [128, 128, 64, 32] if up_dims is None else list(up_dims)
        down_dims = [32, 32, 64, 128] if down_dims is None else list(down_dims)
        out_dim = in_dim if out_dim is None else out_dim
        assert len(down_dims) == len(up_dims), "len(down_dims) != len(up_dims)"
        assert down_dims[0] == up_dims[-1], "down_dims[0] != up_dims[-1]"
        assert up_dims[0] == down_dims[-1], "up_dims[0] != down_dims[-1]"
        if attn_type not in ("linear_cat", "linear_add"):
            raise NotImplementedError()
        time_dim = int(down_dims[0] * 4) if time_dim is None else time_dim
        self.config = dict(in_dim=in_dim, out_dim=out_dim, down_dims=down_dims, up_dims=up_dims, mid_depth=mid_depth,
                           with_time_emb=with_time_emb, time_dim=time_dim, resnet_block_groups=resnet_block_groups,
                           use_convnext=use_convnext, convnext_mult=convnext_mult, attn_type=attn_type,
                           n_label_class=n_label_class, condition_type=condition_type, label_emb_dim=label_emb_dim)

        self.init_conv = nn.Conv2d(in_dim, down_dims[0], 7, padding=3)
        if with_time_emb:
            self.time_mlp = _seq(nn.Identity(), nn.Linear(down_dims[0], time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        else:
            time_dim, self.time_mlp = None, None

        def block(d_in, d_out, td=time_dim):
            if use_convnext:
                return _convnext_params(d_in, d_out, td, convnext_mult)
            return _resnet_params(d_in, d_out, td, resnet_block_groups)

        def attn(d):
            return _attn_params(d, attn_type, label_emb_dim)

        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        skips = []
        for d_in, d_out in zip(down_dims[:-1], down_dims[1:]):
            self.downs.append(nn.ModuleList([block(d_in, d_out), attn(d_out), block(d_out, d_out), attn(d_out),
                                             nn.Conv2d(d_out, d_out, 4, 2, 1)]))
            skips.append(d_out)
        mid = down_dims[-1]
        self.mid_left, self.mid_right = nn.ModuleList(), nn.ModuleList()
        for _ in range(mid_depth - 1):
            self.mid_left.append(block(mid, mid))
            self.mid_right.append(block(mid * 2, mid))
        self.mid_mid = nn.ModuleList([block(mid, mid), attn(mid), block(mid, mid)])
        for u_in, u_out in zip(up_dims[:-1], up_dims[1:]):
            sk = skips.pop()
            self.ups.append(nn.ModuleList([block(u_in + sk, u_in), attn(u_in), nn.ConvTranspose2d(u_in, u_in, 4, 2, 1),
                                           block(u_in + sk, u_out), attn(u_out), block(u_out + sk, u_out), attn(u_out)]))
        self.final_conv = _seq(block(down_dims[0] + up_dims[-1], up_dims[-1], None), nn.Conv2d(up_dims[-1], out_dim, 3, padding=1))

        self.compute_dtype = "fp32"   # "fp32" (parity tier) or "bf16" (throughput tier); not part of the state dict
        self.hip_graph = False        # replay each forward plan as one captured HIP graph (small-batch latency path)
        self._engine = None
        self.eval()

    # ------------------------------------------------------------------ reference API
    def size(self):
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        print(f"Total parameters: {total}")
        print(f"Trainable parameters: {trainable}")

    def set_compute_dtype(self, name):
        assert name in ("fp32", "bf16", "bf16x3"), name     # bf16x3: fp32 tier with split-precision 3x3 convolutions on the bf16 matrix cores
        if name != self.compute_dtype:
            self.compute_dtype, self._engine = name, None
        return self

    def use_hip_graph(self, on=True):
        """Small-batch latency path: the plan of a forward pass (~250 launches through the C ABI, host-bound below U-Net batch ~4) is
        captured once per (batch, size, condition) as a HIP graph and replayed; inputs / output go through static buffers.  Results are
        bit-identical to the eager plan (same kernels, same order)."""
        self.hip_graph = bool(on)
        if self._engine is not None:
            self._engine.hip_graph = self.hip_graph
        return self

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    # DiffSynthSampler._predict passes paired_halves=True with the doubled batch of classifier-free guidance to models that say they take it
    cfg_paired_halves = True

    @torch.no_grad()
    def forward(self, x, time, condition=None, paired_halves=False):
        """paired_halves (keyword, optional — the reference's call is model(x, t, condition)): the caller guarantees that the two halves of
        x and of time are equal (classifier-free guidance: only `condition` differs); the shared prefix is then computed once."""
        if not x.is_cuda:
            raise RuntimeError("diffusynth_amd.ConditionedUnet runs on MI355X only (HIP kernels, no CPU fallback); "
                               "move the model and inputs to 'cuda'")
        if self._engine is None:
            from .engine import UnetEngine
            self._engine = UnetEngine(self, self.compute_dtype)
            self._engine.hip_graph = self.hip_graph
        if paired_halves:
            # the contract: x[:B/2] == x[B/2:] and time[:B/2] == time[B/2:]; violating it is undefined behaviour (the
            # first half's prefix is used for both).  What is free on the host is checked; DS_CHECK_PAIRED=1 also compares on the device (a sync).
            B = x.shape[0]
            if B % 2 or condition is None:
                raise ValueError("paired_halves=True needs an even batch and a condition (the doubled batch of classifier-free guidance)")
            if torch.is_tensor(time) and not time.is_cuda and not torch.equal(time[:B // 2], time[B // 2:]):
                raise ValueError("paired_halves=True but the two halves of `time` differ")
            if os.environ.get("DS_CHECK_PAIRED") == "1":
                if not (torch.equal(x[:B // 2], x[B // 2:]) and torch.equal(time[:B // 2], time[B // 2:])):
                    raise ValueError("paired_halves=True but the two halves of x / time differ (DS_CHECK_PAIRED=1)")
        return self._engine.forward(x, time, condition, paired=paired_halves)


UNet = ConditionedUnet  # the reference's get_diffusion_model names its instance UNet (diffusion.py:367)

PRODUCTION_CONFIG = dict(in_dim=4, down_dims=[96, 96, 192, 384], up_dims=[384, 384, 192, 96],
                         attn_type="linear_add", condition_type="natural_language_prompt", label_emb_dim=512)


def get_diffusion_model(model_Config, load_pretrain=False, model_name=None, device="cuda"):
    """diffusion.py:354-376 equivalent (checkpoint dict key 'model_state_dict')."""
    net = ConditionedUnet(**model_Config)
    print(f"Model intialized, size: {sum(p.numel() for p in net.parameters() if p.requires_grad)}")
    net.to(device)
    if load_pretrain:
        print(f"Loading weights from models/{model_name}_UNet.pth")
        ckpt = torch.load(f"models/{model_name}_UNet.pth", map_location=device, weights_only=True)
        net.load_state_dict(ckpt["model_state_dict"])
    net.eval()
    return net
