"""Oracle (test infrastructure): CPU restatement of the VQ-GAN quantiser and decoder
used after sampling.  Follows model/VQGAN.py:12-27 (norm/activation), :78-146
(VectorQuantizerEMA.forward, eval), :43-75 (VectorQuantizer), :163-272 (UpSample,
ResnetBlock, LinearAttention), :329-400 (Decoder).  Driven by a reference-format
state dict (keys ``_vq_vae.*`` / ``_decoder._layers.N.*``).
"""
import torch
import torch.nn.functional as F

PRODUCTION_CONFIG = dict(  # app.py:32-35
    in_channels=3, hidden_channels=[80, 160], embedding_dim=4, out_channels=3, block_depth=2,
    attn_pos=[80, 160], attn_with_skip=True, num_embeddings=8192, commitment_cost=0.25, decay=0.99,
    norm_type="groupnorm", act_type="swish", num_groups=16)


def decoder_plan(cfg):
    """Layer list of Decoder.__init__ (VQGAN.py:332-387): [(kind, cin, cout)], index = _layers index."""
    hid = list(reversed(cfg["hidden_channels"]))
    attn = cfg.get("attn_pos") or []
    depth = cfg.get("block_depth", 2)
    plan = [("conv1x1", cfg["embedding_dim"], hid[0])]
    cur = hid[0]

    def stage():
        for _ in range(depth - 1):
            if cur in attn:
                plan.append(("attn", cur, cur))
            plan.append(("res", cur, cur))

    stage()
    for nxt in hid[1:]:
        plan.extend([("norm", cur, cur), ("relu", cur, cur), ("up", cur, nxt)])
        cur = nxt
        stage()
    plan.extend([("norm", cur, cur), ("relu", cur, cur), ("up", cur, cur), ("res", cur, cfg["out_channels"])])
    return plan


def _act(x, act_type):
    """VQGAN.py:20-27."""
    return F.relu(x) if act_type == "relu" else x * torch.sigmoid(x)


def _norm(sd, p, x, groups):
    """VQGAN.py:12-17 — GroupNorm eps 1e-6, or (norm_type="batchnorm": the state dict then carries running statistics) BatchNorm2d in
    inference mode (eps 1e-5, the nn.BatchNorm2d default)."""
    if (p + ".running_mean") in sd:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def vq_resblock(sd, p, x, groups, act_type):
    """VQGAN.py:223-244 with temb=None, double_conv=False, nin shortcut when channels change."""
    h = _act(_norm(sd, p + ".norm1", x, groups), act_type)
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def vq_linattn(sd, p, x, heads=1, with_skip=True):
    """VQGAN.py:261-272 — softmax over n on k only; no q softmax, no scale."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + ".to_qkv.weight"])
    hid = qkv.shape[1] // 3
    q, k, v = (t.reshape(b, heads, hid // heads, h * w) for t in qkv.chunk(3, dim=1))
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, hid, h, w)
    out = F.conv2d(out, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])
    if with_skip:
        out = out + F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return out


@torch.no_grad()
def decoder_forward(sd, cfg, q, prefix="_decoder"):
    """VQGAN.py:390-400."""
    x = q
    groups, act = cfg["num_groups"], cfg["act_type"]
    for i, (kind, cin, cout) in enumerate(decoder_plan(cfg)):
        p = f"{prefix}._layers.{i}"
        if kind == "conv1x1":
            x = F.conv2d(x, sd[p + ".weight"])
        elif kind == "attn":
            x = vq_linattn(sd, p, x, 1, cfg.get("attn_with_skip", True))
        elif kind == "res":
            x = vq_resblock(sd, p, x, groups, act)
        elif kind == "norm":
            x = _norm(sd, p, x, groups)
        elif kind == "relu":
            x = F.relu(x)
        elif kind == "up":
            x = F.conv_transpose2d(x, sd[p + "._conv2d.weight"], sd[p + "._conv2d.bias"], stride=2, padding=1)
    return torch.stack([F.softplus(x[:, 0]), torch.tanh(x[:, 1]), torch.tanh(x[:, 2])], dim=1)


@torch.no_grad()
def vq_forward(codebook, z, commitment_cost=0.25, ema=True):
    """VQGAN.py:98-146 in eval mode (ema=False: the non-EMA VectorQuantizer, VQGAN.py:43-75 — the same search, loss = q_latent_loss +
    commitment_cost * e_latent_loss).  Returns (quantized BCHW, loss, perplexity, indices)."""
    zl = z.permute(0, 2, 3, 1).contiguous()
    flat = zl.view(-1, codebook.shape[1])
    dist = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2 * torch.matmul(flat, codebook.t()))
    idx = torch.argmin(dist, dim=1)
    quant = codebook[idx].view(zl.shape)
    loss = commitment_cost * F.mse_loss(quant, zl)
    if not ema:
        loss = F.mse_loss(quant, zl) + loss
    quant = zl + (quant - zl)
    probs = torch.bincount(idx, minlength=codebook.shape[0]).float() / idx.numel()
    perplexity = torch.exp(-torch.sum(probs * torch.log(probs + 1e-10)))
    return quant.permute(0, 3, 1, 2).contiguous(), loss, perplexity, idx


def encoder_plan(cfg):
    """Layer list of Encoder.__init__ (VQGAN.py:278-321): [(kind, cin, cout)], index = _layers index."""
    hid = list(cfg["hidden_channels"])
    attn = cfg.get("attn_pos") or []
    depth = cfg.get("block_depth", 2)
    plan = [("down", cfg["in_channels"], hid[0])]
    cur = hid[0]

    def stage():
        for _ in range(depth - 1):
            plan.append(("res", cur, cur))
            if cur in attn:
                plan.append(("attn", cur, cur))

    for nxt in hid[1:]:
        stage()
        plan.extend([("norm", cur, cur), ("relu", cur, cur), ("down", cur, nxt)])
        cur = nxt
    stage()
    plan.extend([("norm", cur, cur), ("relu", cur, cur), ("conv1x1b", cur, cfg["embedding_dim"])])
    return plan


@torch.no_grad()
def encoder_forward(sd, cfg, x, prefix="_encoder"):
    """VQGAN.py:323-326.  The reference builds the Encoder with the literal act_type="act_type" (VQGAN.py:441), which
    takes the swish branch of `nonlinearity` (VQGAN.py:20-27) regardless of the configured activation."""
    groups = cfg["num_groups"]
    for i, (kind, cin, cout) in enumerate(encoder_plan(cfg)):
        p = f"{prefix}._layers.{i}"
        if kind == "down":
            x = F.conv2d(x, sd[p + "._conv2d.weight"], sd[p + "._conv2d.bias"], stride=2, padding=1)
        elif kind == "attn":
            x = vq_linattn(sd, p, x, 1, cfg.get("attn_with_skip", True))
        elif kind == "res":
            x = vq_resblock(sd, p, x, groups, "swish")
        elif kind == "norm":
            x = _norm(sd, p, x, groups)
        elif kind == "relu":
            x = F.relu(x)
        elif kind == "conv1x1b":
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"])
    return x
