"""GPU parity of the latent -> audio tail (VQ, VQGAN decoder, ISTFT+/iSTFT) against reference goldens / the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_keys, load_golden, rel_err
from diffusynth_amd.synth import synth_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vae(vqgan_sd):
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN
    m = VQGAN(**PRODUCTION_CONFIG)
    m.load_state_dict(vqgan_sd)
    return m.to("cuda")


def test_quantiser_matches_reference(vae):
    g = load_golden("tail")
    z = torch.from_numpy(g["vq_z"]).cuda()
    q, loss, (perp, a, b) = vae._vq_vae(z)
    assert a is None and b is None
    idx = vae._vq_vae.last_indices.flatten().cpu()
    assert (idx == torch.from_numpy(g["vq_idx"])).float().mean().item() > 0.999
    same = (idx == torch.from_numpy(g["vq_idx"])).view(z.shape[0], z.shape[2], z.shape[3])[:, None].expand_as(z)
    assert torch.equal(q.cpu()[same], torch.from_numpy(g["vq_q"])[same])
    assert abs(loss.item() - g["vq_loss"].item()) < 1e-3 * abs(g["vq_loss"].item())
    assert abs(perp.item() - g["vq_perplexity"].item()) < 1e-2 * g["vq_perplexity"].item()


def test_non_ema_quantiser_matches_reference():
    """VQGAN(decay=0)'s VectorQuantizer (VQGAN.py:30-75) on the device against the reference's forward pass (golden/vq_plain.npz)."""
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN, VectorQuantizer
    g = load_golden("vq_plain")
    m = VQGAN(**dict(PRODUCTION_CONFIG, decay=0.0))
    assert type(m._vq_vae) is VectorQuantizer
    m._vq_vae.load_state_dict({"_embedding.weight": torch.from_numpy(g["codebook"])})
    m = m.to("cuda")
    z = torch.from_numpy(g["z"]).cuda()
    q, loss, (perp, a, b) = m._vq_vae(z)
    assert a is None and b is None
    idx = m._vq_vae.last_indices.flatten().cpu()
    same1 = idx == torch.from_numpy(g["idx"])
    assert same1.float().mean().item() > 0.999
    same = same1.view(z.shape[0], z.shape[2], z.shape[3])[:, None].expand_as(z)
    assert torch.equal(q.cpu()[same], torch.from_numpy(g["q"])[same])
    assert abs(loss.item() - g["loss"].item()) < 1e-3 * abs(g["loss"].item())
    assert abs(perp.item() - g["perplexity"].item()) < 1e-2 * g["perplexity"].item()


def test_batchnorm_vqgan_variant_matches_reference():
    """VQGAN(norm_type="batchnorm") (VQGAN.py:15-16) on the device, both tiers: decoder and encoder against the reference's forward passes
    (golden/vq_bn.npz).  The running statistics are folded into a per-channel affine at pack time."""
    import numpy as np
    from diffusynth_amd.synth import synth_state_dict
    from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN
    g = load_golden("vq_bn")
    spec = [(str(k), tuple(int(d) for d in str(s).split(";") if d)) for k, s in zip(g["keys"], g["shapes"])]
    m = VQGAN(**dict(PRODUCTION_CONFIG, norm_type="batchnorm"))
    m.load_state_dict(synth_state_dict(spec))
    m = m.to("cuda")
    for tier, tol in (("fp32", 1e-3), ("bf16", 5e-2)):
        m._decoder.set_compute_dtype(tier)
        m._encoder.set_compute_dtype(tier)
        ey = rel_err(m._decoder(torch.from_numpy(g["dec_q"]).cuda()).cpu(), g["dec_y"])
        ez = rel_err(m._encoder(torch.from_numpy(g["enc_x"]).cuda()).cpu(), g["enc_z"])
        print(f"batchnorm VQGAN {tier}: decoder {ey:.2e} encoder {ez:.2e}")
        assert ey < tol and ez < tol


@pytest.mark.parametrize("name", ["dec", "dec2"])
def test_decoder_fp32_matches_reference(vae, name):
    g = load_golden("tail")
    vae._decoder.set_compute_dtype("fp32")
    y = vae._decoder(torch.from_numpy(g[name + "_q"]).cuda())
    err = rel_err(y.cpu(), g[name + "_y"])
    print(f"decoder fp32 {name}: rel err {err:.2e}")
    assert y.shape == g[name + "_y"].shape and err < 1e-3


def test_decoder_bf16_error_is_bounded(vae):
    g = load_golden("tail")
    vae._decoder.set_compute_dtype("bf16")
    y = vae._decoder(torch.from_numpy(g["dec_q"]).cuda())
    vae._decoder.set_compute_dtype("fp32")
    err = rel_err(y.cpu(), g["dec_y"])
    print(f"decoder bf16: rel err {err:.2e}")
    assert err < 5e-2


def test_dec_final_kernel_matches_torch():
    """ds_dec_final: the decoder's last ResnetBlock(80 -> 3) + softplus / tanh in one pass (VQGAN.py:177-244,390-398) vs torch fp64 on the
    bf16-rounded input: ragged image (tile borders inside the image and at its edge), 16- and 32-wide tile shapes."""
    import ctypes as C
    from diffusynth_amd import _lib as L
    for (B, Hh, Ww) in ((2, 19, 45), (1, 9, 12)):
        g = torch.Generator().manual_seed(Hh)
        Cc, G = 80, 16
        x = (torch.randn(B, Cc, Hh, Ww, generator=g) * 1.7 + 0.3).bfloat16()
        gamma, beta = torch.randn(Cc, generator=g) * 0.3 + 1.0, torch.randn(Cc, generator=g) * 0.2
        w3, b3 = torch.randn(3, Cc, 3, 3, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1
        wn, bn = torch.randn(3, Cc, 1, 1, generator=g) * 0.1, torch.randn(3, generator=g) * 0.1
        xd = x.double()
        hn = F.group_norm(xd, G, gamma.double(), beta.double(), 1e-6)
        y = F.conv2d(hn * torch.sigmoid(hn), w3.bfloat16().double(), b3.double(), padding=1) + F.conv2d(xd, wn.double(), bn.double())
        want = torch.stack([F.softplus(y[:, 0]), torch.tanh(y[:, 1]), torch.tanh(y[:, 2])], 1)
        xg = xd.view(B, G, -1)
        mean, var = xg.mean(2), xg.var(2, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        ab = torch.stack([rstd, rstd * mean], 2).float().cuda().contiguous()
        xn = x.permute(0, 2, 3, 1).contiguous().cuda()
        w3d = w3.float().contiguous().cuda()
        n16 = L.load().ds_pack_conv_elems(96, 3, 3, 16, 0)
        wpk = torch.empty(n16, dtype=torch.bfloat16, device="cuda")
        pp = L.PackConvParams(w=w3d.data_ptr(), gamma=None, dst=wpk.data_ptr(), dtype=L.DS_BF16, Cout=3, Cin=Cc, cin_pad=96, KH=3, KW=3, cout_pad=16,
                              transposed=0, k_order=1)
        L.call("ds_pack_conv_weight", C.byref(pp), L.current_stream())
        gd, bd, b3d, wnd, bnd = gamma.cuda(), beta.cuda(), b3.cuda(), wn.view(3, Cc).contiguous().cuda(), bn.cuda()
        out = torch.full((B, 3, Hh, Ww), float("nan"), device="cuda")
        L.call("ds_dec_final", xn.data_ptr(), B, Hh, Ww, Cc, ab.data_ptr(), G, gd.data_ptr(), bd.data_ptr(), wpk.data_ptr(), b3d.data_ptr(),
               wnd.data_ptr(), bnd.data_ptr(), out.data_ptr(), L.current_stream())
        torch.cuda.synchronize()
        err = rel_err(out.cpu(), want)
        print(f"dec_final {B}x{Hh}x{Ww}: rel err {err:.2e}")
        assert err < 6e-3                      # swish(GroupNorm(x)) is rounded to bf16 before the 3x3 (as in the unfused bf16 path)


@pytest.mark.parametrize("hw,cin", [((8, 32), 80), ((19, 45), 80), ((64, 64), 80), ((4, 32), 160), ((19, 45), 160), ((32, 64), 160)])
def test_convt4x4_c80_matches_torch(hw, cin):
    """ds_convt4x4_c80 (the decoder's last Upsample on its own kernel) == F.conv_transpose2d(x, w, b, stride 2, padding 1) on the bf16-rounded
    operands: the four output phases, ragged tiles and image borders (zeros from the buffer range check), several tiles per block."""
    import hip_helpers as h
    from diffusynth_amd import _lib as L
    B, (Hh, Ww) = 3, hw
    x = synth_input("t_u8_x%s%d" % (hw, cin), (B, cin, Hh, Ww))
    w = synth_input("t_u8_w%d" % cin, (cin, 80, 4, 4), 0.05)
    b = synth_input("t_u8_b", (80,))
    xd = h.to_nhwc(x, L.DS_BF16)
    want = F.conv_transpose2d(h.from_nhwc(xd), w.bfloat16().float(), b, stride=2, padding=1)
    wd, bd = w.contiguous().cuda(), b.cuda()
    wp = torch.empty(L.load().ds_convt4x4_c80_weight_elems(cin), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_convt4x4_c80", wd.data_ptr(), cin, 80, wp.data_ptr(), st)
    out = torch.full((B, 2 * Hh, 2 * Ww, 80), float("nan"), device="cuda").to(torch.bfloat16)
    L.call("ds_convt4x4_c80", xd.data_ptr(), B, Hh, Ww, cin, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), None, 0, None, None, None, st)
    h.sync()
    got = h.from_nhwc(out)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 6e-3, rel_err(got, want)
    # the same layer reading relu(GroupNorm(16, 80)(x)): the affine applied while the halo is staged (zero padding AFTER the activation)
    G = 16
    gamma, beta = synth_input("t_u8_g%d" % cin, (cin,)) * 0.3 + 1.0, synth_input("t_u8_be%d" % cin, (cin,)) * 0.5
    xq = h.from_nhwc(xd)
    xn = F.relu(F.group_norm(xq, G, gamma, beta, eps=1e-6))
    want2 = F.conv_transpose2d(xn, w.bfloat16().float(), b, stride=2, padding=1)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats", xd.data_ptr(), L.DS_BF16, B, Hh * Ww, cin, G, 1e-6, ab.data_ptr(), st)
    out.fill_(float("nan"))
    gd, bed = gamma.cuda(), beta.cuda()
    slots = L.load().ds_convt4x4_c80_stats_slots(B, Hh, Ww, cin)
    ws = torch.full((B, slots, 80, 2), float("nan"), device="cuda")
    L.call("ds_convt4x4_c80", xd.data_ptr(), B, Hh, Ww, cin, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), ab.data_ptr(), G, gd.data_ptr(), bed.data_ptr(),
           ws.data_ptr(), st)
    ab2 = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats_finish", ws.data_ptr(), B, slots, 80, G, 4 * Hh * Ww, 1e-6, ab2.data_ptr(), st)
    h.sync()
    got2 = h.from_nhwc(out)
    assert torch.isfinite(got2).all() and torch.isfinite(ws).all()
    assert rel_err(got2, want2) < 1e-2, rel_err(got2, want2)
    # the statistics partials of the output: (rstd, rstd * mean) per (sample, group) of GroupNorm(16, 80) over the result
    gr = want2.reshape(B, G, -1).double()
    rstd = 1.0 / torch.sqrt(gr.var(dim=2, unbiased=False) + 1e-6)
    assert (ab2[:, :, 0].cpu().double() - rstd).abs().max() / rstd.abs().max() < 5e-3
    assert (ab2[:, :, 1].cpu().double() - rstd * gr.mean(dim=2)).abs().max() < 2e-2


@pytest.mark.parametrize("hw,act", [((4, 32), "silu"), ((19, 45), "silu"), ((64, 64), "relu"), ((9, 70), None)])
def test_conv3x3_c80_matches_torch(hw, act):
    """ds_conv3x3_c80 == x + conv3x3(act(GroupNorm(16, 80)(x))) + bias on the bf16-rounded operands (act None: plain convolution without the
    norm and the residual): ragged tiles, borders (zero padding AFTER the activation), several tiles per block, sample changes inside a run."""
    import hip_helpers as h
    from diffusynth_amd import _lib as L
    B, (Hh, Ww), G = 3, hw, 16
    x = synth_input("t_c8_x%s" % (hw,), (B, 80, Hh, Ww)) * 1.5 + 0.2
    w = synth_input("t_c8_w", (80, 80, 3, 3), 0.04)
    b = synth_input("t_c8_b", (80,))
    gamma, beta = synth_input("t_c8_g", (80,)) * 0.3 + 1.0, synth_input("t_c8_be", (80,)) * 0.5
    xd = h.to_nhwc(x, L.DS_BF16)
    xq = h.from_nhwc(xd)
    wq = w.bfloat16().float()
    if act is None:
        want = F.conv2d(xq, wq, b, padding=1)
    else:
        xn = F.group_norm(xq, G, gamma, beta, eps=1e-6)
        want = xq + F.conv2d(F.silu(xn) if act == "silu" else F.relu(xn), wq, b, padding=1)
    wd, bd, gd, bed = w.contiguous().cuda(), b.cuda(), gamma.cuda(), beta.cuda()
    wp = torch.empty(L.load().ds_conv3x3_c80_weight_elems(), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_conv3x3_c80", wd.data_ptr(), 80, 80, wp.data_ptr(), st)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats", xd.data_ptr(), L.DS_BF16, B, Hh * Ww, 80, G, 1e-6, ab.data_ptr(), st)
    out = torch.full((B, Hh, Ww, 80), float("nan"), device="cuda").to(torch.bfloat16)
    if act is None:
        L.call("ds_conv3x3_c80", xd.data_ptr(), B, Hh, Ww, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), None, 0, None, None, L.ACT_NONE, 0, None, st)
        h.sync()
    else:
        slots = L.load().ds_conv3x3_c80_stats_slots(B, Hh, Ww)
        ws = torch.full((B, slots, 80, 2), float("nan"), device="cuda")
        L.call("ds_conv3x3_c80", xd.data_ptr(), B, Hh, Ww, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), ab.data_ptr(), G, gd.data_ptr(),
               bed.data_ptr(), L.ACT_SILU if act == "silu" else L.ACT_RELU, 1, ws.data_ptr(), st)
        ab2 = torch.empty(B, G, 2, device="cuda")
        L.call("ds_gn_stats_finish", ws.data_ptr(), B, slots, 80, G, Hh * Ww, 1e-6, ab2.data_ptr(), st)
        h.sync()
        assert torch.isfinite(ws).all()
        gr = want.reshape(B, G, -1).double()
        rstd = 1.0 / torch.sqrt(gr.var(dim=2, unbiased=False) + 1e-6)
        assert (ab2[:, :, 0].cpu().double() - rstd).abs().max() / rstd.abs().max() < 5e-3
        assert (ab2[:, :, 1].cpu().double() - rstd * gr.mean(dim=2)).abs().max() < 2e-2
    got = h.from_nhwc(out)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 1e-2, rel_err(got, want)
    if act is not None:
        # the same block as two launches (what the decoder plan runs since r04): h = act(GroupNorm(x)) by ds_gn_apply, out = x + conv3x3(h) + bias
        import ctypes as C
        hact = torch.full((B, Hh, Ww, 80), float("nan"), device="cuda").to(torch.bfloat16)
        gp = L.GnApplyParams(x=xd.data_ptr(), res=None, out=hact.data_ptr(), gn_ab=ab.data_ptr(), gamma=gd.data_ptr(), beta=bed.data_ptr(), cbias=None,
                             cb_stride=0, B=B, HW=Hh * Ww, C=80, G=G, act=L.ACT_SILU if act == "silu" else L.ACT_RELU, dtype=L.DS_BF16)
        L.call("ds_gn_apply", C.byref(gp), st)
        out2 = torch.full((B, Hh, Ww, 80), float("nan"), device="cuda").to(torch.bfloat16)
        ws2 = torch.full((B, slots, 80, 2), float("nan"), device="cuda")
        L.call("ds_conv3x3_c80_res", hact.data_ptr(), xd.data_ptr(), B, Hh, Ww, wp.data_ptr(), bd.data_ptr(), out2.data_ptr(), ws2.data_ptr(), st)
        h.sync()
        got2 = h.from_nhwc(out2)
        assert torch.isfinite(got2).all() and torch.isfinite(ws2).all()
        assert rel_err(got2, want) < 1e-2 and rel_err(got2, got) < 1e-2, (rel_err(got2, want), rel_err(got2, got))


@pytest.mark.parametrize("dim,hw,B,skip", [(80, (19, 45), 3, True), (80, (64, 64), 2, True), (160, (9, 70), 2, True), (160, (32, 64), 3, False), (80, (4, 8), 1, False)])
def test_vq_attn_fused_matches_torch(dim, hw, B, skip):
    """ds_vq_attn_context + ds_vq_attn_output == LinearAttention(dim, 1, 32) (VQGAN.py:246-272) on the bf16-rounded x and to_qkv weights:
    ragged groups / tiles (N not a multiple of 128 or 32), images smaller than one group, several blocks per sample, with and without the
    nin_shortcut; the per-channel statistics slots against the GroupNorm(16) statistics of the stored output."""
    import ctypes as C
    import hip_helpers as h
    from diffusynth_amd import _lib as L
    Hh, Ww = hw
    N, G = Hh * Ww, 16
    x = synth_input("t_va_x%s%d" % (hw, dim), (B, dim, Hh, Ww)) * 1.3 + 0.1
    wqkv = synth_input("t_va_wqkv%d" % dim, (96, dim)) * (2.0 / dim ** 0.5)
    wout = synth_input("t_va_wo%d" % dim, (dim, 32)) * 0.2
    bout = synth_input("t_va_bo%d" % dim, (dim,)) * 0.3
    wnin = synth_input("t_va_wn%d" % dim, (dim, dim)) * (1.0 / dim ** 0.5) if skip else None
    bnin = synth_input("t_va_bn%d" % dim, (dim,)) * 0.3
    xd = h.to_nhwc(x, L.DS_BF16)
    xq = h.from_nhwc(xd).double().reshape(B, dim, N)
    wq = wqkv.bfloat16().double()
    q, k, v = torch.einsum("oc,bcn->bon", wq, xq).chunk(3, dim=1)
    ctx = torch.einsum("bdn,ben->bde", k.softmax(dim=-1), v)
    out = torch.einsum("bde,bdn->ben", ctx, q)
    want = torch.einsum("ce,ben->bcn", wout.double(), out) + bout.double()[None, :, None]
    if skip:
        want = want + torch.einsum("co,bon->bcn", wnin.double(), xq) + bnin.double()[None, :, None]
    lib = L.load()
    st = L.current_stream()
    nseg = lib.ds_vq_attn_segments(B, N, dim)
    part = torch.full((lib.ds_linattn_part_floats(B, 1, nseg),), float("nan"), device="cuda")
    cx = torch.full((B, 32, 32), float("nan"), device="cuda")
    wfold = torch.empty(lib.ds_vq_attn_wfold_bytes(B, dim), dtype=torch.uint8, device="cuda")
    y = torch.full((B, Hh, Ww, dim), float("nan"), device="cuda").to(torch.bfloat16)
    ws = torch.full((B, nseg // 4, dim, 2), float("nan"), device="cuda")
    wqkv_d = wqkv.bfloat16().contiguous().cuda()
    wq_d, wout_d = wqkv[:32].contiguous().cuda(), wout.contiguous().cuda()
    wnin_d = wnin.contiguous().cuda() if skip else None
    bias_d = (bout + bnin if skip else bout).contiguous().cuda()
    p = L.VqAttnParams(x=xd.data_ptr(), B=B, N=N, C=dim, nseg=nseg, wqkv=wqkv_d.data_ptr(), wq=wq_d.data_ptr(), wout=wout_d.data_ptr(),
                       wnin=L.ptr(wnin_d), bias=bias_d.data_ptr(), part=part.data_ptr(), ctx=cx.data_ptr(), wfold=wfold.data_ptr(), y=y.data_ptr(),
                       stats_ws=ws.data_ptr())
    L.call("ds_vq_attn_context", C.byref(p), st)
    L.call("ds_vq_attn_output", C.byref(p), st)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats_finish", ws.data_ptr(), B, nseg // 4, dim, G, N, 1e-6, ab.data_ptr(), st)
    h.sync()
    assert rel_err(cx.cpu().double(), ctx) < 6e-3, rel_err(cx.cpu().double(), ctx)      # k, v from bf16 operands, P / V rounded to bf16 for ctx
    got = h.from_nhwc(y).double().reshape(B, dim, N)
    assert torch.isfinite(got).all()
    err = rel_err(got, want)
    print(f"vq_attn {dim} {hw} B={B}: rel err {err:.2e}")
    assert err < 1e-2, err
    assert torch.isfinite(ws).all()
    gr = got.reshape(B, G, -1)
    rstd = 1.0 / torch.sqrt(gr.var(dim=2, unbiased=False) + 1e-6)
    assert (ab[:, :, 0].cpu().double() - rstd).abs().max() / rstd.abs().max() < 1e-4
    assert (ab[:, :, 1].cpu().double() - rstd * gr.mean(dim=2)).abs().max() < 1e-3


def test_decoder_fused_attention_matches_unfused(vae):
    """The bf16 decoder with its LinearAttention blocks on csrc/vq_attn.hip against the same decoder on the unfused chain (DS_NO_VQ_ATTN=1: to_qkv,
    context, output, merged to_out | nin_shortcut): two bf16 evaluations of the same network."""
    import os
    g = load_golden("tail")
    q = torch.from_numpy(g["dec_q"]).cuda()
    vae._decoder.set_compute_dtype("bf16")
    y1 = vae._decoder(q)
    os.environ["DS_NO_VQ_ATTN"] = "1"
    try:
        vae._decoder.set_compute_dtype("fp32")
        vae._decoder.set_compute_dtype("bf16")          # rebuilds the engine (packing reads the switch)
        y0 = vae._decoder(q)
    finally:
        del os.environ["DS_NO_VQ_ATTN"]
        vae._decoder.set_compute_dtype("fp32")
    err = rel_err(y1.cpu(), y0.cpu())
    print(f"decoder bf16, fused vs unfused attention: rel err {err:.2e}; vs fp32 reference {rel_err(y1.cpu(), g['dec_y']):.2e} / {rel_err(y0.cpu(), g['dec_y']):.2e}")
    assert err < 2e-2 and rel_err(y1.cpu(), g["dec_y"]) < 5e-2


@pytest.mark.parametrize("cin,cout,hw,B", [(4, 160, (9, 7), 3), (8, 80, (4, 32), 2), (4, 8, (1, 5), 1)])
def test_conv1x1_in_nchw_matches_torch(cin, cout, hw, B):
    """ds_conv1x1_in_nchw: the decoder's first layer (VQGAN.py:345, Conv2d(embedding_dim, hidden, 1)) on the NCHW fp32 latent -> bf16 NHWC, with and
    without a bias; pixel counts that are not a multiple of the rows a block walks."""
    import hip_helpers as h
    from diffusynth_amd import _lib as L
    Hh, Ww = hw
    x = synth_input("t_ic_x%d%s" % (cin, hw), (B, cin, Hh, Ww)) * 1.4
    w = synth_input("t_ic_w%d_%d" % (cin, cout), (cout, cin)) * 0.5
    b = synth_input("t_ic_b%d" % cout, (cout,))
    xd, wd, bd = x.contiguous().cuda(), w.contiguous().cuda(), b.cuda()
    for bias in (None, bd):
        want = F.conv2d(x.double(), w.double().view(cout, cin, 1, 1), b.double() if bias is not None else None)
        out = torch.full((B, Hh, Ww, cout), float("nan"), device="cuda").to(torch.bfloat16)
        L.call("ds_conv1x1_in_nchw", xd.data_ptr(), B, cin, Hh * Ww, wd.data_ptr(), L.ptr(bias), cout, out.data_ptr(), L.current_stream())
        h.sync()
        got = h.from_nhwc(out).double()
        assert torch.isfinite(got).all() and rel_err(got, want) < 5e-3, rel_err(got, want)      # fp32 arithmetic, bf16 store


def test_vq_stats_matches_torch():
    """ds_vq_stats: mean((q - z)^2), perplexity and the module's loss from (z, q, idx) against the reference's expressions (VQGAN.py:62-73, 131-144),
    both quantiser flavours; indices from a skewed distribution (unused codes: p log(p + 1e-10) = 0)."""
    from diffusynth_amd import _lib as L
    B, D, Hh, Ww, K = 3, 4, 11, 13, 512
    z = synth_input("t_vs_z", (B, D, Hh, Ww))
    q = z + 0.1 * synth_input("t_vs_q", (B, D, Hh, Ww))
    g = torch.Generator().manual_seed(5)
    idx = (torch.rand(B * Hh * Ww, generator=g) ** 3 * 300).long()
    mse = F.mse_loss(q.double(), z.double())
    pr = torch.bincount(idx, minlength=K).double() / idx.numel()
    perp = torch.exp(-torch.sum(pr * torch.log(pr + 1e-10)))
    zd, qd, idd = z.contiguous().cuda(), q.contiguous().cuda(), idx.cuda()
    ws = torch.empty(L.load().ds_vq_stats_ws_bytes(K), dtype=torch.uint8, device="cuda")
    for cc, ema, want_loss in ((0.25, 1, 0.25 * mse), (0.4, 0, mse + 0.4 * mse)):
        out3 = torch.full((3,), float("nan"), device="cuda")
        L.call("ds_vq_stats", zd.data_ptr(), qd.data_ptr(), idd.data_ptr(), B, D, Hh * Ww, K, cc, ema, out3.data_ptr(), ws.data_ptr(), L.current_stream())
        got = out3.cpu().double()
        assert abs(got[0] - mse) / mse < 1e-5 and abs(got[1] - perp) / perp < 1e-4 and abs(got[2] - want_loss) / want_loss < 1e-5, (got, mse, perp)


def test_decoder_upsample_kernel_matches_generic(vae):
    """The decoder (bf16) with its 80-channel block and last Upsample on their own kernels (ds_conv3x3_c80, ds_convt4x4_c80) against the same
    decoder with those layers on the generic kernels (DS_NO_UP80=1, DS_NO_C80=1)."""
    import os
    q = synth_input("t_u8_q", (2, 4, 32, 16)).cuda()
    dec = vae._decoder
    dec.set_compute_dtype("bf16")
    try:
        y_new = dec(q)
        os.environ["DS_NO_UP80"] = "1"
        os.environ["DS_NO_C80"] = "1"
        dec.set_compute_dtype("fp32")
        dec.set_compute_dtype("bf16")                 # (re-pack: the switch is read when the layers are packed)
        y_old = dec(q)
    finally:
        os.environ.pop("DS_NO_UP80", None)
        os.environ.pop("DS_NO_C80", None)
        dec.set_compute_dtype("fp32")
    assert torch.isfinite(y_new).all()
    assert (y_new - y_old).abs().max().item() < 3e-2 * y_old.abs().max().item()


def test_decoder_fused_final_block_matches_unfused(vae):
    """The bf16 decoder with its last block on ds_dec_final vs the same decoder with DS_NO_DEC_FINAL=1 semantics (a second engine built
    with the switch on): both are bf16 evaluations of the same network — they agree to bf16 rounding."""
    import os
    g = load_golden("tail")
    q = torch.from_numpy(g["dec_q"]).cuda()
    vae._decoder.set_compute_dtype("bf16")
    y1 = vae._decoder(q)
    os.environ["DS_NO_DEC_FINAL"] = "1"
    try:
        vae._decoder.set_compute_dtype("fp32")
        vae._decoder.set_compute_dtype("bf16")          # rebuilds the engine (packing reads the switch)
        y0 = vae._decoder(q)
    finally:
        del os.environ["DS_NO_DEC_FINAL"]
        vae._decoder.set_compute_dtype("fp32")
    err = rel_err(y1.cpu(), y0.cpu())
    print(f"decoder bf16, fused vs unfused last block: rel err {err:.2e}")
    assert err < 2e-2 and rel_err(y1.cpu(), g["dec_y"]) < 5e-2


def test_latents_to_audio_matches_oracle(vae, vqgan_sd):
    """Config-5 style end-to-end tail from IDENTICAL quantised latents: decoder + ISTFT+ + iSTFT vs the CPU oracle."""
    from diffusynth_amd.vocoder import encodeBatch2GradioOutput_STFT, latents_to_audio
    from oracle import vocoder_ref as V
    from oracle import vqgan_ref as Q
    q = synth_input("tail_e2e_q", (2, 4, 128, 3))          # decoder output (2,3,512,12): F=512 like production
    vae._decoder.set_compute_dtype("fp32")
    audio = latents_to_audio(vae._decoder, q.cuda())
    dec_ref = Q.decoder_forward(vqgan_sd, Q.PRODUCTION_CONFIG, q)
    ref = np.stack(V.latents_to_audio(dec_ref.numpy()))
    assert audio.shape == ref.shape == (2, 256 * 11)
    err = rel_err(audio.cpu(), ref)
    print(f"latents->audio fp32: rel err {err:.2e}")
    assert err < 1e-3
    out = encodeBatch2GradioOutput_STFT(vae._decoder, q.numpy())
    assert len(out) == 6 and out[2][0].dtype == np.float64 and rel_err(np.stack(out[2]), ref) < 1e-3


@pytest.mark.parametrize("name", ["enc", "enc2"])
def test_encoder_fp32_matches_reference(vae, name):
    """VQGAN encoder on device vs the reference's own outputs."""
    g = load_golden("front")
    vae._encoder.set_compute_dtype("fp32")
    z = vae._encoder(torch.from_numpy(g[name + "_x"]).cuda())
    err = rel_err(z.cpu(), g[name + "_z"])
    print(f"encoder fp32 {name}: rel err {err:.2e}")
    assert z.shape == g[name + "_z"].shape and err < 1e-3
    vae._encoder.set_compute_dtype("bf16")
    zb = vae._encoder(torch.from_numpy(g[name + "_x"]).cuda())
    vae._encoder.set_compute_dtype("fp32")
    assert rel_err(zb.cpu(), g[name + "_z"]) < 5e-2


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_stft_plus_and_audio_round_trip(vae, pad_mode):
    """STFT+ kernel vs the oracle (librosa semantics, parity unpinned) and the STFT -> iSTFT round trip, which is a
    size-independent property: a signal analysed and re-synthesised with the same window/hop comes back unchanged."""
    from diffusynth_amd.vocoder import InputBatch2Encode_STFT, audio_to_stft_representation, stft_representation_to_audio
    from oracle import vocoder_ref as V
    y = synth_input("front_audio2", (2, 256 * 40))
    enc = audio_to_stft_representation(y.cuda(), time_resolution=48, pad_mode=pad_mode)
    assert enc.shape == (2, 3, 512, 48)
    ref = np.stack([V.encode_stft(V.pad_stft(V.stft(s.numpy(), pad_mode=pad_mode), 48)) for s in y])
    assert rel_err(enc[:, 0].cpu(), ref[:, 0]) < 1e-4                       # log-magnitude
    strong = torch.from_numpy(ref[:, 0]) > 0.05                            # phase is ill-conditioned where |X| ~ 0
    assert (enc[:, 1:].cpu() - torch.from_numpy(ref[:, 1:]))[strong[:, None].expand(-1, 2, -1, -1)].abs().max().item() < 1e-2
    assert torch.equal(enc[:, :, :, 41:].cpu(), torch.tensor([0.0, 1.0, 0.0]).view(1, 3, 1, 1).expand(2, 3, 512, 7))
    back = stft_representation_to_audio(enc[:, :, :, :41].contiguous())
    assert back.shape == (2, 256 * 40)
    # DC is dropped by pad_STFT (the representation has no bin 0): compare against the DC-free interior of the signal
    D = np.stack([V.stft(s.numpy(), pad_mode=pad_mode) for s in y])
    D[:, 0, :] = 0
    want = np.stack([V.istft(d.astype(np.complex128), 256, 1024) for d in D])
    assert rel_err(back.cpu()[:, 1024:-1024], want[:, 1024:-1024]) < 1e-3
    z = InputBatch2Encode_STFT(vae._encoder, enc, quantizer=vae._vq_vae)
    assert z[3].shape == (2, 4, 128, 12) and z[4].shape == (2, 4, 128, 12)


def test_istft_accepts_a_misaligned_view():
    """ds_istft_plus on a contiguous view whose storage offset is not a multiple of 4 floats (the 16-byte gather path must
    not be taken on it): same audio as the aligned copy, bit for bit."""
    from diffusynth_amd import _lib as L
    from diffusynth_amd.vocoder import stft_representation_to_audio
    B, F, T = 2, 512, 12
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(B, 3, F, T, generator=g)
    enc[:, 0] = enc[:, 0].abs() * 0.5
    want = stft_representation_to_audio(enc.cuda())
    flat = torch.zeros(enc.numel() + 1, device="cuda")
    flat[1:] = enc.flatten().cuda()
    view = flat[1:].view(B, 3, F, T)
    assert view.data_ptr() % 16 == 4 and view.is_contiguous()
    ws = torch.empty(L.load().ds_istft_ws_floats(B, F, T), device="cuda")
    got = torch.empty(B, 256 * (T - 1), device="cuda")
    L.call("ds_istft_plus", view.data_ptr(), B, F, T, 256, ws.data_ptr(), got.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_istft_phase_of_tiny_and_huge_cos_sin_pairs():
    """ds_istft_plus takes arbitrary (cos, sin) channels: the phase is atan2(sin, cos) (tools.py:334-345), which is scale-free.  Pairs whose
    squares underflow (1e-23: c^2 + s^2 == 0 in fp32), denormal pairs and pairs whose squares overflow (1e25) must give the audio of the
    same phases at unit scale — the kernel normalises (cos, sin) by |(cos, sin)| and used to feed c^2 + s^2 straight to v_rsq_f32."""
    from diffusynth_amd.vocoder import stft_representation_to_audio
    from oracle import vocoder_ref as V
    B, F, T = 2, 512, 9
    g = torch.Generator().manual_seed(11)
    ph = (torch.rand(B, F, T, generator=g) * 2 - 1) * 3.14159
    mag = torch.rand(B, F, T, generator=g) * 0.5
    enc = torch.stack([mag, torch.cos(ph), torch.sin(ph)], 1).contiguous()
    want = stft_representation_to_audio(enc.cuda()).cpu()
    ref = np.stack([V.istft(V.depad_stft(V.decode_stft(e.double().numpy())), 256, 1024) for e in enc])
    assert rel_err(want, ref) < 1e-4
    for scale in (1e-23, 1e-40, 1e25):
        e2 = enc.clone()
        e2[:, 1:] = (enc[:, 1:].double() * scale).float()
        got = stft_representation_to_audio(e2.cuda()).cpu()
        assert torch.isfinite(got).all(), scale
        # (at 1e-40 the fp32 pair itself is a few-bit denormal: its phase IS coarser — compare with what numpy's atan2 makes of the same fp32 inputs)
        ref2 = np.stack([V.istft(V.depad_stft(V.decode_stft(e.double().numpy())), 256, 1024) for e in e2])
        assert rel_err(got, ref2) < 1e-4, scale
    # atan2(0, 0) = 0: a zero pair keeps the magnitude on the real axis
    e3 = enc.clone()
    e3[:, 1:, 5:9] = 0.0
    got = stft_representation_to_audio(e3.cuda()).cpu()
    ref3 = np.stack([V.istft(V.depad_stft(V.decode_stft(e.double().numpy())), 256, 1024) for e in e3])
    assert rel_err(got, ref3) < 1e-4


def test_codebook_rewritten_through_data_is_seen(vae):
    """`weight.data.normal_()` is the reference's own idiom for setting the codebook (VQGAN.py:38, :92) and does not bump the parameter's
    version counter: the quantiser must not answer from a codebook it cached on an earlier forward."""
    vq = vae._vq_vae
    w = vq._embedding.weight
    keep = w.data.clone()
    z = synth_input("vq_rewrite_z", (1, w.shape[1], 16, 8)).cuda()
    try:
        q0 = vq(z)[0].clone()
        g = torch.Generator().manual_seed(5)
        w.data.copy_(torch.randn(w.shape, generator=g).to(w.device) * 0.7)
        q1 = vq(z)[0]
        cb = w.detach().float()
        flat = z.permute(0, 2, 3, 1).reshape(-1, w.shape[1])
        idx = torch.cdist(flat.double(), cb.double()).argmin(1)
        want = cb[idx].view(1, 16, 8, -1).permute(0, 3, 1, 2)
        assert not torch.equal(q0, q1)
        # the kernel's answer must be rows of the NEW codebook, and the nearest ones (near-ties may resolve differently in its split-precision
        # distances than in float64: > 99.9 % equal indices, like test_non_ema_quantiser_matches_reference)
        idx_k = vq.last_indices.flatten()
        e_k = cb[idx_k].view(1, 16, 8, -1).permute(0, 3, 1, 2)
        assert torch.allclose(q1, z + (e_k - z), rtol=0, atol=1e-6)         # (straight-through form of VQGAN.py:66 / :136: inputs + (quantized - inputs))
        assert (idx_k == idx).float().mean().item() > 0.999
        assert (q1 - want).abs().max().item() < 0.5                       # (a near-tie picks a neighbouring code, never a far one)
    finally:
        w.data.copy_(keep)


def test_text_condition_head_matches_reference():
    """ProjectionHead on device (ds_linear x2 + ds_add_layernorm per layer) vs the reference's outputs,
    with the reference's state-dict names."""
    from diffusynth_amd.synth import synth_state_dict
    from diffusynth_amd.text_head import ProjectionHead
    g = load_golden("head")
    for tag, (din, dout, nl) in {"h1": (512, 512, 1), "h2": (768, 512, 2)}.items():
        head = ProjectionHead(din, dout, 0.1, num_layers=nl)
        spec = [(tag + "." + k, tuple(v.shape)) for k, v in head.state_dict().items()]
        head.load_state_dict({k[len(tag) + 1:]: v for k, v in synth_state_dict(spec).items()})
        head.cuda()
        y = head(torch.from_numpy(g[tag + "_x"]).cuda())
        err = rel_err(y.cpu(), g[tag + "_y"])
        print(f"text head {tag}: rel err {err:.2e}")
        assert y.shape == g[tag + "_y"].shape and err < 1e-4
    with pytest.raises(RuntimeError):
        head(torch.zeros(2, 768))          # no CPU fallback
