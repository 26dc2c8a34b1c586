"""Oracle (test infrastructure): functional CPU restatement of the conditional
U-Net noise predictor, driven directly by a reference-format state dict.

Follows model/diffusion.py:21-258 (graph), model/diffusion_components.py:22-293
(blocks).  No nn.Module: every function takes the flat ``sd`` mapping
``reference key -> tensor`` plus a key prefix, so the key names double as the
citation of which reference parameter is used where.
"""
import math

import torch
import torch.nn.functional as F

PRODUCTION_CONFIG = dict(  # app.py:40
    in_dim=4, down_dims=[96, 96, 192, 384], up_dims=[384, 384, 192, 96],
    attn_type="linear_add", condition_type="natural_language_prompt", label_emb_dim=512)


def full_config(cfg):
    """Fill the constructor defaults of model/diffusion.py:22-38."""
    c = dict(out_dim=None, down_dims=None, up_dims=None, mid_depth=3, with_time_emb=True, time_dim=None,
             resnet_block_groups=8, use_convnext=True, convnext_mult=2, attn_type="linear_cat",
             n_label_class=11, condition_type="instrument_family", label_emb_dim=128)
    c.update(cfg)
    if c["up_dims"] is None:
        c["up_dims"] = [128, 128, 64, 32]
    if c["down_dims"] is None:
        c["down_dims"] = [32, 32, 64, 128]
    if c["out_dim"] is None:
        c["out_dim"] = c["in_dim"]
    if c["time_dim"] is None:
        c["time_dim"] = int(c["down_dims"][0] * 4)
    return c


# ----------------------------------------------------------------------------- primitives

def sinusoid(time, dim):
    """components:42-56 — [sin(t f_i), cos(t f_i)], f_i = exp(-i ln(1e4)/(dim/2-1))."""
    half = dim // 2
    freqs = torch.exp(torch.arange(half, device=time.device) * -(math.log(10000) / (half - 1)))
    arg = time[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def _conv(sd, p, x, **kw):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), **kw)


def _gn(sd, p, x, groups, eps=1e-5):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def convnext_block(sd, p, x, temb):
    """components:107-139."""
    h = F.conv2d(x, sd[p + ".ds_conv.weight"], sd[p + ".ds_conv.bias"], padding=3, groups=x.shape[1])
    if (p + ".mlp.1.weight") in sd and temb is not None:
        h = h + _lin(sd, p + ".mlp.1", F.gelu(temb))[:, :, None, None]
    if (p + ".net.0.weight") in sd:
        h = _gn(sd, p + ".net.0", h, 1)
    h = _conv(sd, p + ".net.1", h, padding=1)
    h = F.gelu(h)
    h = _gn(sd, p + ".net.3", h, 1)
    h = _conv(sd, p + ".net.4", h, padding=1)
    res = _conv(sd, p + ".res_conv", x) if (p + ".res_conv.weight") in sd else x
    return h + res


def resnet_block(sd, p, x, temb, groups):
    """components:59-104 (the scale_shift argument is never supplied: :96)."""
    h = F.silu(_gn(sd, p + ".block1.norm", _conv(sd, p + ".block1.proj", x, padding=1), groups))
    if (p + ".mlp.1.weight") in sd and temb is not None:
        h = _lin(sd, p + ".mlp.1", F.silu(temb))[:, :, None, None] + h
    h = F.silu(_gn(sd, p + ".block2.norm", _conv(sd, p + ".block2.proj", h, padding=1), groups))
    res = _conv(sd, p + ".res_conv", x) if (p + ".res_conv.weight") in sd else x
    return h + res


def _split_heads(t, heads):
    b, c, h, w = t.shape
    return t.reshape(b, heads, c // heads, h * w)


def linear_attention(sd, p, x, cond, kind, heads=4, dim_head=32):
    """components:171-207 (kind='linear_cat') and :252-293 (kind='linear_add').
    ``p`` is the prefix of the attention module itself (…fn.fn)."""
    b, c, hh, ww = x.shape
    q, k, v = (_split_heads(t, heads) for t in _conv(sd, p + ".to_qkv", x).chunk(3, dim=1))
    if cond is not None:
        if kind == "linear_add":
            k = k + _lin(sd, p + ".label_key", cond).view(b, heads, dim_head, 1)
            q = q + _lin(sd, p + ".label_query", cond).view(b, heads, dim_head, 1)
        else:
            k = torch.cat([k, _lin(sd, p + ".label_key", cond).view(b, heads, dim_head, 1)], dim=-1)
            v = torch.cat([v, _lin(sd, p + ".label_value", cond).view(b, heads, dim_head, 1)], dim=-1)
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dim_head, hh, ww)
    out = _conv(sd, p + ".to_out.0", out)
    return _gn(sd, p + ".to_out.1", out, 1)


def attn_block(sd, p, x, cond, kind):
    """Residual(PreNorm(dim, attn)) — components:22-29,142-152."""
    y = _gn(sd, p + ".fn.norm", x, 1)
    return linear_attention(sd, p + ".fn.fn", y, cond, kind) + x


def pad_and_concat(enc, dec):
    """components:210-249 — zero-pad decoder map to encoder size (left/top = delta//2), encoder channels first."""
    dw = enc.shape[3] - dec.shape[3]
    dh = enc.shape[2] - dec.shape[2]
    dec = F.pad(dec, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))
    return torch.cat((enc, dec), dim=1)


# ----------------------------------------------------------------------------- graph

def unet_forward(sd, cfg, x, time, condition=None, taps=None):
    """model/diffusion.py:187-258.  ``taps`` (optional dict) receives named intermediates."""
    c = full_config(cfg)
    nlev = len(c["down_dims"]) - 1
    kind = c["attn_type"]
    if c["use_convnext"]:
        def block(p, t, temb):
            return convnext_block(sd, p, t, temb)
    else:
        def block(p, t, temb):
            return resnet_block(sd, p, t, temb, c["resnet_block_groups"])

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    cond = None
    if condition is not None:
        if c["condition_type"] == "natural_language_prompt":
            cond = _lin(sd, "label_embedding.embedding", condition)
        else:
            cond = F.embedding(condition, sd["label_embedding.embedding.weight"])

    skips = []
    x = _conv(sd, "init_conv", x, padding=3)
    tap("init_conv", x)
    skips.append(x)
    temb = None
    if c["with_time_emb"]:
        temb = sinusoid(time, c["down_dims"][0])
        temb = _lin(sd, "time_mlp.3", F.gelu(_lin(sd, "time_mlp.1", temb)))
        tap("time_emb", temb)

    for i in range(nlev):
        x = block(f"downs.{i}.0", x, temb)
        tap(f"downs.{i}.0", x)
        x = attn_block(sd, f"downs.{i}.1", x, cond, kind)
        tap(f"downs.{i}.1", x)
        skips.append(x)
        x = block(f"downs.{i}.2", x, temb)
        x = attn_block(sd, f"downs.{i}.3", x, cond, kind)
        skips.append(x)
        x = _conv(sd, f"downs.{i}.4", x, stride=2, padding=1)
        tap(f"downs.{i}.4", x)
        skips.append(x)

    for j in range(c["mid_depth"] - 1):
        x = block(f"mid_left.{j}", x, temb)
        skips.append(x)
    x = block("mid_mid.0", x, temb)
    x = attn_block(sd, "mid_mid.1", x, cond, kind)
    x = block("mid_mid.2", x, temb)
    tap("mid_mid", x)
    for j in range(c["mid_depth"] - 1):
        x = block(f"mid_right.{j}", pad_and_concat(skips.pop(), x), temb)

    for i in range(nlev):
        x = block(f"ups.{i}.0", pad_and_concat(skips.pop(), x), temb)
        x = attn_block(sd, f"ups.{i}.1", x, cond, kind)
        x = F.conv_transpose2d(x, sd[f"ups.{i}.2.weight"], sd[f"ups.{i}.2.bias"], stride=2, padding=1)
        tap(f"ups.{i}.2", x)
        x = block(f"ups.{i}.3", pad_and_concat(skips.pop(), x), temb)
        x = attn_block(sd, f"ups.{i}.4", x, cond, kind)
        x = block(f"ups.{i}.5", pad_and_concat(skips.pop(), x), temb)
        x = attn_block(sd, f"ups.{i}.6", x, cond, kind)
        tap(f"ups.{i}.6", x)

    x = block("final_conv.0", pad_and_concat(skips.pop(), x), None)
    return _conv(sd, "final_conv.1", x, padding=1)


class RefUnet:
    """Callable wrapper with the reference's ``model(x, t, condition)`` duck type."""

    def __init__(self, sd, cfg=None):
        self.sd = {k: v.float() if v.is_floating_point() else v for k, v in sd.items()}
        self.cfg = dict(PRODUCTION_CONFIG if cfg is None else cfg)

    @torch.no_grad()
    def __call__(self, x, time, condition=None):
        return unet_forward(self.sd, self.cfg, x, time, condition)
