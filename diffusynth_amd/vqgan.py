"""VQGAN — drop-in for the parts of model/VQGAN.py used after sampling:

    vae._vq_vae(z)  -> (quantized BCHW, loss, (perplexity, None, None))     VQGAN.py:98-146 (eval)
    vae._decoder(q) -> (B, 3, 4H, 4W)  [softplus | tanh | tanh]             VQGAN.py:329-400

with the reference's state-dict names and shapes (74 tensors incl. the encoder, whose parameters
run too: ``vae._encoder(stft_plus)`` is row 2 of SURVEY §8f, the audio -> latent front end).  The
quantiser is one nearest-code kernel (no N x 8192 distance matrix, no one-hot matmul); the decoder
is a plan over the same HIP kernels as the U-Net (implicit-GEMM convolutions incl. the transposed
4x4, linear attention, GroupNorm(16) + swish / ReLU passes) plus the tail activation kernel.
"""
import ctypes as C
import os

import torch
from torch import nn

from . import _lib as L
from .engine import _EngineBase, _PlanBuilder
from .unet import _Holder

PRODUCTION_CONFIG = dict(in_channels=3, hidden_channels=[80, 160], embedding_dim=4, out_channels=3, block_depth=2,
                         attn_pos=[80, 160], attn_with_skip=True, num_embeddings=8192, commitment_cost=0.25, decay=0.99,
                         norm_type="groupnorm", act_type="swish", num_groups=16)


def _norm(ch, groups, norm_type="groupnorm"):
    """VQGAN.py:12-17."""
    if norm_type == "batchnorm":
        return nn.BatchNorm2d(ch)
    return nn.GroupNorm(num_groups=groups, num_channels=ch, eps=1e-6, affine=True)


def _res_params(cin, cout, groups, norm_type="groupnorm"):
    h = _Holder(norm1=_norm(cin, groups, norm_type), conv1=nn.Conv2d(cin, cout, 3, 1, 1), temb_proj=nn.Linear(512, cout))
    if cin != cout:
        h.nin_shortcut = nn.Conv2d(cin, cout, 1)
    return h


def _attn_params(dim, with_skip):
    h = _Holder(to_qkv=nn.Conv2d(dim, 96, 1, bias=False), to_out=nn.Conv2d(32, dim, 1))
    if with_skip:
        h.nin_shortcut = nn.Conv2d(dim, dim, 1)
    return h


def _layer_plan(cfg, decoder):
    """[(kind, cin, cout)] in _layers order — Decoder.__init__ (VQGAN.py:332-387) / Encoder.__init__ (:278-321)."""
    hid = list(cfg["hidden_channels"])
    attn = cfg.get("attn_pos") or []
    depth = cfg.get("block_depth", 2)
    plan = []

    def stage(cur, attn_first):
        for _ in range(depth - 1):
            if attn_first:
                if cur in attn:
                    plan.append(("attn", cur, cur))
                plan.append(("res", cur, cur))
            else:
                plan.append(("res", cur, cur))
                if cur in attn:
                    plan.append(("attn", cur, cur))

    if decoder:
        hid = hid[::-1]
        cur = hid[0]
        plan.append(("conv1x1", cfg["embedding_dim"], cur))
        stage(cur, True)
        for nxt in hid[1:]:
            plan += [("norm", cur, cur), ("relu", cur, cur), ("up", cur, nxt)]
            cur = nxt
            stage(cur, True)
        plan += [("norm", cur, cur), ("relu", cur, cur), ("up", cur, cur), ("res", cur, cfg["out_channels"])]
    else:
        cur = hid[0]
        plan.append(("down", cfg["in_channels"], cur))
        for nxt in hid[1:]:
            stage(cur, False)
            plan += [("norm", cur, cur), ("relu", cur, cur), ("down", cur, nxt)]
            cur = nxt
        stage(cur, False)
        plan += [("norm", cur, cur), ("relu", cur, cur), ("conv1x1b", cur, cfg["embedding_dim"])]
    return plan


def _build_layers(cfg, decoder):
    g, nt = cfg["num_groups"], cfg.get("norm_type", "groupnorm")
    mods = []
    for kind, cin, cout in _layer_plan(cfg, decoder):
        if kind == "conv1x1":
            mods.append(nn.Conv2d(cin, cout, 1, bias=False))
        elif kind == "conv1x1b":
            mods.append(nn.Conv2d(cin, cout, 1))
        elif kind == "attn":
            mods.append(_attn_params(cin, cfg.get("attn_with_skip", True)))
        elif kind == "res":
            mods.append(_res_params(cin, cout, g, nt))
        elif kind == "norm":
            mods.append(_norm(cin, g, nt))
        elif kind == "relu":
            mods.append(nn.ReLU())
        elif kind == "up":
            mods.append(_Holder(_conv2d=nn.ConvTranspose2d(cin, cout, 4, 2, 1)))
        elif kind == "down":
            mods.append(_Holder(_conv2d=nn.Conv2d(cin, cout, 4, 2, 1)))
    return nn.ModuleList(mods)


class _NearestCode(nn.Module):
    """Eval-mode forward shared by both quantisers of the reference (VQGAN.py:43-75 and :98-146 are the same nearest-code search; they
    differ in the codebook's initialisation, in the EMA state of the training-time update and in the loss they return)."""

    def _search(self, inputs, ema):
        if self.training:
            raise RuntimeError("diffusynth_amd quantisers are inference-only (codebook updates are out of scope)")
        if not inputs.is_cuda:
            raise RuntimeError("diffusynth_amd quantiser runs on MI355X only (ds_vq_nearest); no CPU fallback")
        z = inputs.contiguous().float()
        B, D, H, W = z.shape
        w = self._embedding.weight
        # the codebook is read in place (an fp32 contiguous parameter aliases; anything else is converted per call) and |e|^2 is re-derived on
        # every forward — two tiny launches, no sync.  A cache keyed on (data_ptr, _version) went stale under `weight.data.normal_()`, the
        # reference's own idiom for setting the codebook (VQGAN.py:38, :92): writes through .data do not bump _version.
        cb = w.detach()
        if cb.dtype != torch.float32 or not cb.is_contiguous():
            cb = cb.float().contiguous()
        esq = torch.sum(cb ** 2, dim=1).contiguous()                 # same expression as VQGAN.py:49 / :108
        q = torch.empty_like(z)
        idx = torch.empty(B * H * W, dtype=torch.int64, device=z.device)
        L.call("ds_vq_nearest", z.data_ptr(), cb.data_ptr(), esq.data_ptr(), B, D, H * W, cb.shape[0], q.data_ptr(), idx.data_ptr(),
               L.current_stream())
        # mse = mean((q - z)^2), perplexity = exp(-sum p log(p + 1e-10)) (callers discard both: text2sound.py:128) — one statistics kernel over
        # (z, q, idx) instead of a dozen torch launches and the host synchronisation inside torch.bincount
        out3 = torch.empty(3, dtype=torch.float32, device=z.device)
        ws = torch.empty(L.load().ds_vq_stats_ws_bytes(cb.shape[0]), dtype=torch.uint8, device=z.device)
        L.call("ds_vq_stats", z.data_ptr(), q.data_ptr(), idx.data_ptr(), B, D, H * W, cb.shape[0], float(self._commitment_cost), 1 if ema else 0,
               out3.data_ptr(), ws.data_ptr(), L.current_stream())
        self.last_indices = idx.view(B, H, W)
        return q, out3[2], out3[1]


class VectorQuantizer(_NearestCode):
    """VQGAN.py:30-75 (chosen by VQGAN when decay == 0, :441-446): codebook initialised uniform(-1/K, 1/K), state dict = {_embedding.weight},
    loss = q_latent_loss + commitment_cost * e_latent_loss (numerically (1 + commitment_cost) * mse in the forward pass)."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.uniform_(-1 / num_embeddings, 1 / num_embeddings)
        self._commitment_cost = commitment_cost
        self.eval()

    @torch.no_grad()
    def forward(self, inputs):
        q, loss, perplexity = self._search(inputs, ema=False)
        return q, loss, (perplexity, None, None)


class VectorQuantizerEMA(_NearestCode):
    """Eval-mode EMA quantiser (VQGAN.py:78-146): codebook initialised normal(), state dict = {_embedding.weight, _ema_cluster_size, _ema_w},
    loss = commitment_cost * e_latent_loss."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay, epsilon=1e-5):
        super().__init__()
        self._embedding_dim, self._num_embeddings = embedding_dim, num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.normal_()
        self._commitment_cost = commitment_cost
        self.register_buffer("_ema_cluster_size", torch.zeros(num_embeddings))
        self._ema_w = nn.Parameter(torch.randn(num_embeddings, embedding_dim))
        self._decay, self._epsilon = decay, epsilon
        self.eval()

    @torch.no_grad()
    def forward(self, inputs):
        q, loss, perplexity = self._search(inputs, ema=True)
        return q, loss, (perplexity, None, None)


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._layers = _build_layers(cfg, decoder=True)
        self.compute_dtype = "fp32"
        self._engine = None
        self.eval()

    def set_compute_dtype(self, name):
        assert name in ("fp32", "bf16"), name
        if name != self.compute_dtype:
            self.compute_dtype, self._engine = name, None
        return self

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("diffusynth_amd.Decoder runs on MI355X only (HIP kernels, no CPU fallback)")
        if self._engine is None:
            self._engine = DecoderEngine(self, self.compute_dtype)
        return self._engine.forward(x)


class Encoder(nn.Module):
    """VQGAN.Encoder (VQGAN.py:275-326): (B, 3, 512, T) STFT+ representation -> (B, embedding_dim, 128, T/4) latent.
    Like the reference (VQGAN.py:441 passes the literal act_type="act_type") its ResnetBlocks always use swish."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._layers = _build_layers(cfg, decoder=False)
        self.compute_dtype = "fp32"
        self._engine = None
        self.eval()

    def set_compute_dtype(self, name):
        assert name in ("fp32", "bf16"), name
        if name != self.compute_dtype:
            self.compute_dtype, self._engine = name, None
        return self

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("diffusynth_amd.Encoder runs on MI355X only (HIP kernels, no CPU fallback)")
        if self._engine is None:
            self._engine = DecoderEngine(self, self.compute_dtype, decoder=False)
        return self._engine.forward(x)


class VQGAN(nn.Module):
    """Constructor = model/VQGAN.py:435-451."""

    def __init__(self, in_channels, hidden_channels, embedding_dim, out_channels, block_depth=2, attn_pos=None,
                 attn_with_skip=True, norm_type="groupnorm", act_type="relu", num_embeddings=1024, commitment_cost=0.25,
                 decay=0.99, num_groups=32):
        super().__init__()
        if norm_type not in ("groupnorm", "batchnorm"):
            raise NotImplementedError(f"norm_type={norm_type!r}: VQGAN.py:12-17 knows 'batchnorm' and GroupNorm")
        cfg = dict(norm_type=norm_type, in_channels=in_channels, hidden_channels=list(hidden_channels), embedding_dim=embedding_dim,
                   out_channels=out_channels, block_depth=block_depth, attn_pos=list(attn_pos or []),
                   attn_with_skip=attn_with_skip, act_type=act_type, num_embeddings=num_embeddings,
                   commitment_cost=commitment_cost, decay=decay, num_groups=num_groups)
        self.config = cfg
        self._encoder = Encoder(cfg)
        # VQGAN.py:441-446
        if decay > 0.0:
            self._vq_vae = VectorQuantizerEMA(num_embeddings, embedding_dim, commitment_cost, decay)
        else:
            self._vq_vae = VectorQuantizer(num_embeddings, embedding_dim, commitment_cost)
        self._decoder = Decoder(cfg)
        self.eval()

    def load_state_dict(self, *a, **k):
        self._decoder._engine = self._encoder._engine = None
        return super().load_state_dict(*a, **k)


def get_VQGAN(model_Config, load_pretrain=False, model_name=None, device="cuda"):
    """VQGAN.py:564-586 equivalent."""
    net = VQGAN(**model_Config)
    net.to(device)
    if load_pretrain:
        ckpt = torch.load(f"models/{model_name}_imageVQVAE.pth", map_location=device, weights_only=True)
        net.load_state_dict(ckpt["model_state_dict"])
    net.eval()
    return net


# =============================================================================== decoder engine
class DecoderEngine(_EngineBase):
    """Plan executor for the VQGAN Decoder and (decoder=False) Encoder layer stacks."""

    def __init__(self, module, compute_dtype, decoder=True):
        self._init_common(module, compute_dtype)
        self.cfg = module.cfg
        self.is_decoder = decoder
        # norm_type="batchnorm" (VQGAN.py:15-16), inference: a per-channel affine from the running statistics — folded at pack time and applied by
        # the GroupNorm apply kernel with unit factors (no statistics pass; the fused GroupNorm kernels of the 80-channel stage are not used)
        self.bn = self.cfg.get("norm_type", "groupnorm") == "batchnorm"
        self._bn_ab = {}
        self.plan_list = _layer_plan(self.cfg, decoder)
        cin = self.cfg["embedding_dim"] if decoder else self.cfg["in_channels"]
        self.in_ch = cin
        self.out_ch = 3 if decoder else self.cfg["embedding_dim"]
        self.cin0 = (cin + self.vec - 1) // self.vec * self.vec
        with torch.cuda.device(self.dev):
            self.P = [self._pack_layer(kind, layer, i) for i, ((kind, _, _), layer) in enumerate(zip(self.plan_list, module._layers))]
            self._pack_done()

    def _norm_params(self, nm):
        if not self.bn:
            return (self._f32(nm.weight), self._f32(nm.bias))
        scale = self._f32(nm.weight) / torch.sqrt(self._f32(nm.running_var) + nm.eps)
        return (scale.contiguous(), (self._f32(nm.bias) - self._f32(nm.running_mean) * scale).contiguous())

    def bn_unit_ab(self, B):
        """[B][2] = (rstd, rstd * mean) = (1, 0): ds_gn_apply then computes act(x * gamma' + beta')."""
        t = self._bn_ab.get(B)
        if t is None:
            t = self._bn_ab[B] = torch.tensor([[1.0, 0.0]] * B, dtype=torch.float32, device=self.dev)
        return t

    def _pack_layer(self, kind, m, i):
        if kind == "conv1x1":
            d = {"conv": self._pack_conv(m.weight, None, cin_pad=self.cin0 if i == 0 else None), "in_nchw": None}
            cout, cin = int(m.weight.shape[0]), int(m.weight.shape[1])
            if i == 0 and self.dt == L.DS_BF16 and cin in (4, 8) and cout % 8 == 0 and os.environ.get("DS_NO_IN_CONV", "0") != "1":
                # layout change + 1x1 convolution of the NCHW latent in one pass (ds_conv1x1_in_nchw)
                d["in_nchw"] = self._f32(m.weight).reshape(cout, cin).contiguous()
            return d
        if kind == "attn":
            d = {"qkv": self._pack_conv(m.to_qkv.weight, None), "out": self._pack_conv(m.to_out.weight, m.to_out.bias), "nin": None}
            dim = int(m.to_qkv.weight.shape[1])
            if self.dt == L.DS_BF16 and dim in (80, 160) and tuple(m.to_qkv.weight.shape[:2]) == (96, dim) and os.environ.get("DS_NO_VQ_ATTN", "0") != "1":
                # the whole block on csrc/vq_attn.hip: context from x, then ONE per-sample 1x1 convolution (q enters linearly)
                wqkv = self._f32(m.to_qkv.weight).reshape(96, dim)
                bias = self._f32(m.to_out.bias)
                wnin = None
                if hasattr(m, "nin_shortcut"):
                    wnin = self._f32(m.nin_shortcut.weight).reshape(dim, dim).contiguous()
                    bias = bias + self._f32(m.nin_shortcut.bias)
                d["fused"] = {"wqkv": wqkv.to(torch.bfloat16).contiguous(), "wq": wqkv[:32].contiguous(),
                              "wout": self._f32(m.to_out.weight).reshape(dim, 32).contiguous(), "wnin": wnin, "bias": bias.contiguous()}
            if hasattr(m, "nin_shortcut"):
                d["nin"] = self._pack_conv(m.nin_shortcut.weight, m.nin_shortcut.bias)
                if os.environ.get("DS_NO_ATTN_MERGE", "0") != "1":
                    # nin_shortcut(x) + to_out(a) = ONE 1x1 convolution over the channel concat (x | a) with the weights side by side: one
                    # launch, and the dim-channel tensor is written once instead of written, re-read and re-written
                    wm = torch.cat([m.nin_shortcut.weight.detach().float(), m.to_out.weight.detach().float()], 1)
                    bm = m.nin_shortcut.bias.detach().float() + m.to_out.bias.detach().float()
                    d["merged"] = self._pack_conv(wm, bm)
            return d
        if kind == "res":
            small = m.conv1.weight.shape[0] < 8
            d = {"norm": self._norm_params(m.norm1),
                 "conv": self._pack_conv(m.conv1.weight, m.conv1.bias, small_out=small, halo=not small), "nin": None}
            if hasattr(m, "nin_shortcut"):
                d["nin"] = self._pack_conv(m.nin_shortcut.weight, m.nin_shortcut.bias, small_out=small)
            cout, cin = m.conv1.weight.shape[:2]
            d["c80"] = None
            if (self.dt == L.DS_BF16 and cout == 80 and cin == 80 and d["nin"] is None and not self.bn and os.environ.get("DS_NO_C80", "0") != "1"):
                # the whole 80-channel block body as one kernel (conv3x3_c80.hip): GroupNorm + activation on load, residual in the epilogue
                wf = self._f32(m.conv1.weight)
                wp = torch.empty(L.load().ds_conv3x3_c80_weight_elems(), dtype=torch.bfloat16, device=self.dev)
                L.call("ds_pack_conv3x3_c80", wf.data_ptr(), 80, 80, wp.data_ptr(), L.current_stream())
                self._pack_tmp.append(wf)
                d["c80"] = (wp, self._f32(m.conv1.bias))
            if (small and cout == 3 and self.is_decoder and self.dt == L.DS_BF16 and d["nin"] is not None and cin % 8 == 0 and cin <= 96
                    and not self.bn and os.environ.get("DS_NO_DEC_FINAL", "0") != "1"):
                # the decoder's last block + output activations as one kernel (dec_final.hip): 3x3 weight as 16-row chunk-major tiles, Cin padded to 96
                w = self._f32(m.conv1.weight)
                n16 = L.load().ds_pack_conv_elems(96, 3, 3, 16, 0)
                w3 = torch.empty(n16, dtype=torch.bfloat16, device=self.dev)
                pp = L.PackConvParams(w=w.data_ptr(), gamma=None, dst=w3.data_ptr(), dtype=L.DS_BF16, Cout=3, Cin=cin, cin_pad=96, KH=3, KW=3,
                                      cout_pad=16, transposed=0, k_order=1)
                L.call("ds_pack_conv_weight", C.byref(pp), L.current_stream())
                self._pack_tmp.append(w)
                d["final"] = (w3, self._f32(m.conv1.bias), self._f32(m.nin_shortcut.weight).reshape(3, cin).contiguous(), self._f32(m.nin_shortcut.bias))
            return d
        if kind == "norm":
            return {"norm": self._norm_params(m)}
        if kind == "up":
            d = {"conv": self._pack_conv(m._conv2d.weight, m._conv2d.bias, transposed=True), "up80": None}
            w = m._conv2d.weight
            if (self.dt == L.DS_BF16 and tuple(w.shape) in ((80, 80, 4, 4), (160, 80, 4, 4)) and not self.bn and os.environ.get("DS_NO_UP80", "0") != "1"):
                # ConvTranspose2d(80 | 160, 80, 4, 2, 1) on its own kernel (convt4x4_c80.hip): K steps of two (tap, 16-channel group) pairs
                wf = self._f32(w)
                cin = int(w.shape[0])
                wp = torch.empty(L.load().ds_convt4x4_c80_weight_elems(cin), dtype=torch.bfloat16, device=self.dev)
                L.call("ds_pack_convt4x4_c80", wf.data_ptr(), cin, 80, wp.data_ptr(), L.current_stream())
                self._pack_tmp.append(wf)
                d["up80"] = (wp, self._f32(m._conv2d.bias) if m._conv2d.bias is not None else None)
            return d
        if kind == "down":
            return {"conv": self._pack_conv(m._conv2d.weight, m._conv2d.bias, cin_pad=self.cin0 if i == 0 else None)}
        if kind == "conv1x1b":
            return {"conv": self._pack_conv(m.weight, m.bias, small_out=True)}
        return {}

    def forward(self, q):
        B, Cq, H, W = q.shape
        assert Cq == self.in_ch, f"expected {self.in_ch} input channels, got {Cq}"
        q = q.float().contiguous()
        with torch.cuda.device(q.device):
            plan = self._cached_plan((B, H, W), lambda: _DecoderPlan(self, B, H, W))
            out = torch.empty((B, self.out_ch, plan.out_hw[0], plan.out_hw[1]), dtype=torch.float32, device=q.device)
            plan.run(q, out)
        return out


class _DecoderPlan(_PlanBuilder):
    def __init__(self, eng, B, H, W):
        super().__init__(eng, B, H, W, False)
        self.tb_all = self.lab_all = None

    def vq_attention(self, d, x):
        """VQGAN.py:261-272: one head of 32, softmax over n on k only, 1x1 skip."""
        e, B = self.e, self.B
        N = x.H * x.W
        f = d.get("fused")
        if f is not None and x.C in (80, 160):
            lib = self.lib
            nseg = lib.ds_vq_attn_segments(B, N, x.C)
            part = self.raw(lib.ds_linattn_part_floats(B, 1, nseg) * 4)
            ctx = self.raw(B * 1024 * 4)
            wfold = self.raw(lib.ds_vq_attn_wfold_bytes(B, x.C))
            y = self.act(x.C, x.H, x.W)
            ws = self.raw(B * (nseg // 4) * x.C * 2 * 4)
            p = L.VqAttnParams(x=x.off, B=B, N=N, C=x.C, nseg=nseg, wqkv=f["wqkv"].data_ptr(), wq=f["wq"].data_ptr(), wout=f["wout"].data_ptr(),
                               wnin=L.ptr(f["wnin"]), bias=f["bias"].data_ptr(), part=part[0], ctx=ctx[0], wfold=wfold[0], y=y.off, stats_ws=ws[0])
            self.conv_meta[len(self.ops) + 1] = (0, 2.0 * B * N * x.C * x.C, f"attention {x.C} @{x.H}x{x.W}: per-sample 1x1")
            self.op("ds_vq_attn_context", p)
            self.op("ds_vq_attn_output", p)
            self.free_raw(part)
            self.free_raw(ctx)
            self.free_raw(wfold)
            y.stats = (ws, nseg // 4, "chan_ws")          # per-channel partial sums of y: the next Normalize only finishes them
            return y
        qkv = self.conv(d["qkv"], x)
        nseg = max(1, min(N // 1024, 16))
        part = self.raw(self.lib.ds_linattn_part_floats(B, 1, nseg) * 4)
        ctx = self.raw(B * 1024 * 4)
        ao = self.act(32, x.H, x.W)
        p = L.AttnParams(qkv=qkv.off, B=B, N=N, heads=1, dtype=e.dt, nseg=nseg, part=part[0], ctx=ctx[0], label_q=None,
                         label_k=None, label_v=None, lq_stride=0, lk_stride=0, lv_stride=0, q_softmax=0, scale=1.0, out=ao.off)
        self.op("ds_linattn_context", p)
        self.op("ds_linattn_output", p)
        self.free(qkv)
        self.free_raw(part)
        self.free_raw(ctx)
        if d.get("merged") is not None:
            out = self.conv(d["merged"], x, src1=ao)
        elif d["nin"] is not None:
            out = self.conv(d["nin"], x)
            out = self.conv(d["out"], ao, res=out, out=out)
        else:
            out = self.conv(d["out"], ao)
        self.free(ao)
        return out

    def _normalize(self, x, nrm, act):
        """act(Normalize(x)) (VQGAN.py:12-27): GroupNorm(num_groups, eps 1e-6) = statistics + apply; BatchNorm2d (inference) = the folded
        per-channel affine through the same apply kernel with unit factors."""
        e = self.e
        if not e.bn:
            return self._gn_explicit(x, nrm, e.cfg["num_groups"], act, eps=1e-6)
        out = self.act(x.C, x.H, x.W)
        p = L.GnApplyParams(x=x.off, res=None, out=out.off, gn_ab=e.bn_unit_ab(self.B).data_ptr(), gamma=nrm[0].data_ptr(), beta=nrm[1].data_ptr(),
                            cbias=None, cb_stride=0, B=self.B, HW=x.H * x.W, C=x.C, G=1, act=act, dtype=e.dt)
        self.op("ds_gn_apply", p)
        return out

    def vq_res(self, d, x):
        """VQGAN.py:223-244 with temb=None: x (or nin_shortcut(x)) + conv3x3(act(GroupNorm(x)))."""
        e = self.e
        # the Encoder's blocks are built with act_type="act_type" (VQGAN.py:441) => swish whatever the config says
        act = L.ACT_RELU if (e.cfg["act_type"] == "relu" and e.is_decoder) else L.ACT_SILU
        if d.get("c80") is not None and x.C == 80:
            # statistics, then ONE kernel: act(GroupNorm(x)) while the halo is staged, 3x3, + bias + x
            B, G = self.B, e.cfg["num_groups"]
            ab = self.raw(B * G * 2 * 4)
            self._stats_op(x, G, 1e-6, ab)
            wp, bias = d["c80"]
            out = self.act(80, x.H, x.W)
            self.conv_meta[len(self.ops)] = (16, 2.0 * B * x.H * x.W * 80 * 9 * 80, f"3x3 80->80 @{x.H}x{x.W}")
            slots = self.lib.ds_conv3x3_c80_stats_slots(B, x.H, x.W)      # per-channel statistics of the output: the next Normalize reads them
            ws = self.raw(B * slots * 80 * 2 * 4)
            if os.environ.get("DS_C80_FUSED_ACT", "0") == "1":
                # (A/B: the norm + activation applied while the kernel stages its halo — one launch, but bound by that arithmetic)
                self.op("ds_conv3x3_c80", x.off, B, x.H, x.W, wp.data_ptr(), bias.data_ptr(), out.off, ab[0], G, d["norm"][0].data_ptr(),
                        d["norm"][1].data_ptr(), act, 1, ws[0])
            else:
                hact = self.act(80, x.H, x.W)
                gp = L.GnApplyParams(x=x.off, res=None, out=hact.off, gn_ab=ab[0], gamma=d["norm"][0].data_ptr(), beta=d["norm"][1].data_ptr(),
                                     cbias=None, cb_stride=0, B=B, HW=x.H * x.W, C=80, G=G, act=act, dtype=e.dt)
                self.op("ds_gn_apply", gp)
                self.op("ds_conv3x3_c80_res", hact.off, x.off, B, x.H, x.W, wp.data_ptr(), bias.data_ptr(), out.off, ws[0])
                self.free(hact)
            self.free_raw(ab)
            out.stats = (ws, slots, "chan_ws")          # (released with the tensor if no Normalize consumes it)
            return out
        h = self._normalize(x, d["norm"], act)
        if d["nin"] is not None:
            out = self.conv(d["nin"], x)
            out = self.conv(d["conv"], h, pad=1, res=out, out=out)
        else:
            out = self.conv(d["conv"], h, pad=1, res=x)
        self.free(h)
        return out

    def build(self, base):
        self.base = base
        e, B, H, W = self.e, self.B, self.H, self.W
        first = e.P[0].get("in_nchw") if (e.plan_list and e.plan_list[0][0] == "conv1x1") else None
        if first is not None:
            x = self.act(e.plan_list[0][2], H, W)
            self.ops.append(("input_conv", x.off, first))
        else:
            xin = self.act(e.cin0, H, W)
            self.ops.append(("input", xin.off))
            x = xin
        pending_norm = None
        fused_gn = None                                # (ab, norm): Normalize + ReLU left to the following layer's input staging
        for idx, ((kind, cin, cout), d) in enumerate(zip(e.plan_list, e.P)):
            if kind == "conv1x1" and idx == 0 and first is not None:
                continue                                   # (done by the input op)
            if kind == "conv1x1":
                y = self.conv(d["conv"], x)
            elif kind == "attn":
                y = self.vq_attention(d, x)
            elif kind == "res" and d.get("final") is not None and x.W > 8 and (kind, cin, cout) == e.plan_list[-1][:3] and d is e.P[-1]:
                # last block of the decoder: GroupNorm statistics, then ONE kernel reads x once and writes the activated fp32 planes
                G = e.cfg["num_groups"]
                ab = self.raw(B * G * 2 * 4)
                self._stats_op(x, G, 1e-6, ab)
                self.ops.append(("final", x.off, x.C, x.H, x.W, ab[0], d))
                self.free_raw(ab)
                self.out_hw = (x.H, x.W)
                self.free(x)
                return
            elif kind == "res":
                y = self.vq_res(d, x)
            elif kind == "norm":
                pending_norm = d["norm"]
                continue
            elif kind == "relu":
                nxt = e.plan_list[idx + 1][0] if idx + 1 < len(e.plan_list) else None
                if nxt == "up" and e.P[idx + 1].get("up80") is not None and x.C in (80, 160):
                    # the 80-channel Upsample applies Normalize + ReLU to its input while staging it: statistics only, no apply pass
                    G = e.cfg["num_groups"]
                    ab = self.raw(B * G * 2 * 4)
                    self._stats_op(x, G, 1e-6, ab)
                    fused_gn = (ab, pending_norm, G)
                    pending_norm = None
                    continue
                y = self._normalize(x, pending_norm, L.ACT_RELU)   # Normalize + nn.ReLU fused
                pending_norm = None
            elif kind == "up" and d.get("up80") is not None and x.C in (80, 160):
                wp, bias = d["up80"]
                y = self.act(80, 2 * x.H, 2 * x.W)
                self.conv_meta[len(self.ops)] = (15, 2.0 * B * x.H * x.W * 4 * 80 * 4 * x.C, f"2x2T {x.C}->80 @{x.H}x{x.W}")
                # per-channel statistics of the output where the next layer is a Normalize (the last Upsample: the final block follows)
                want_ws = idx + 1 < len(e.plan_list) and e.plan_list[idx + 1][0] == "res" and e.P[idx + 1].get("final") is not None
                slots = self.lib.ds_convt4x4_c80_stats_slots(B, x.H, x.W, x.C)
                ws = self.raw(B * slots * 80 * 2 * 4) if want_ws else None
                if fused_gn is not None:
                    ab, nrm, G = fused_gn
                    self.op("ds_convt4x4_c80", x.off, B, x.H, x.W, x.C, wp.data_ptr(), L.ptr(bias), y.off, ab[0], G, nrm[0].data_ptr(), nrm[1].data_ptr(),
                            ws[0] if ws else None)
                    self.free_raw(ab)
                    fused_gn = None
                else:
                    self.op("ds_convt4x4_c80", x.off, B, x.H, x.W, x.C, wp.data_ptr(), L.ptr(bias), y.off, None, 0, None, None, ws[0] if ws else None)
                if ws:
                    y.stats = (ws, slots, "chan_ws")
            elif kind == "up":
                y = self.conv(d["conv"], x)
            elif kind == "down":
                y = self.conv(d["conv"], x, stride=2, pad=1)
            elif kind == "conv1x1b":
                y = self.conv(d["conv"], x)
            else:
                raise NotImplementedError(kind)
            self.free(x)
            x = y
        self.out_hw = (x.H, x.W)
        self.ops.append(("tail" if e.is_decoder else "latent", x.off, x.C))
        self.free(x)

    def run(self, q, out):
        e, B = self.e, self.B
        st = L.current_stream()
        lib = self.lib
        prof = getattr(self, "prof", None)                    # diagnostics (tools/tail_bench.py --ops): an event after every op
        if prof is not None:
            # ... and one in front of the first: whatever the caller enqueued before this plan (the VQ search) is not op 0's time
            self.prof_start = torch.cuda.Event(enable_timing=True)
            self.prof_start.record()
        for k, item in enumerate(self.ops):
            tag = item[0]
            if tag == "input":
                rc = lib.ds_nchw_to_nhwc(q.data_ptr(), B, q.shape[1], self.H, self.W, item[1], e.cin0, e.dt, st)
                name = "ds_nchw_to_nhwc"
            elif tag == "input_conv":
                w = item[2]
                rc = lib.ds_conv1x1_in_nchw(q.data_ptr(), B, q.shape[1], self.H * self.W, w.data_ptr(), None, w.shape[0], item[1], st)
                name = "ds_conv1x1_in_nchw"
            elif tag == "tail":
                rc = lib.ds_decoder_tail(item[1], e.dt, B, item[2], self.out_hw[0] * self.out_hw[1], out.data_ptr(), st)
                name = "ds_decoder_tail"
            elif tag == "final":
                _, xoff, xc, xh, xw, ab, d = item
                w3, b3, wn, bn = d["final"]
                rc = lib.ds_dec_final(xoff, B, xh, xw, xc, ab, e.cfg["num_groups"], d["norm"][0].data_ptr(), d["norm"][1].data_ptr(),
                                      w3.data_ptr(), b3.data_ptr(), wn.data_ptr(), bn.data_ptr(), out.data_ptr(), st)
                name = "ds_dec_final"
            elif tag == "latent":
                rc = lib.ds_nhwc_to_nchw(item[1], e.dt, B, out.shape[1], item[2], self.out_hw[0], self.out_hw[1], out.data_ptr(), st)
                name = "ds_nhwc_to_nchw"
            else:
                fn, args, name = item
                rc = fn(C.byref(args[0]), st) if isinstance(args[0], C.Structure) else fn(*args, st)
            if rc != 0:
                L.check(rc, name)
            if prof is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                prof.append((k, name, ev))
