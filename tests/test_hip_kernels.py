"""GPU parity tests, one kernel class at a time, through the C ABI.
Checker = torch CPU fp32 ops of the same operator (and oracle/ functions for the fused blocks).
Tolerances (max|diff|/max|ref|): fp32 kernels 2e-5 (fp32 MFMA is an exact fma chain; only the
summation order differs), bf16 kernels 2e-2 (8-bit mantissa inputs), sampler step bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from diffusynth_amd import _lib as L
from diffusynth_amd.synth import synth_input

pytestmark = pytest.mark.gpu

TOL = {L.DS_F32: 2e-5, L.DS_BF16: 2e-2}
DTS = [L.DS_F32, L.DS_BF16]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    L.load()


def H():
    import hip_helpers
    return hip_helpers


# ----------------------------------------------------------------------------------------- convolution
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile,cout", [(L.TILE_128x192, 192), (L.TILE_256x96, 96), (L.TILE_64x192, 384), (L.TILE_128x32, 4)])
def test_conv3x3_plain(dt, tile, cout):
    h = H()
    x = synth_input("k_c3_x", (2, 64, 12, 20))
    w = synth_input("k_c3_w%d" % cout, (cout, 64, 3, 3), 0.05)
    b = synth_input("k_c3_b%d" % cout, (cout,))
    pc = h.PackedConv(w, b, dt, tile)
    y, _ = h.run_conv(pc, h.to_nhwc(x, dt), pad=1)
    assert rel_err(h.from_nhwc(y), F.conv2d(x, w, b, padding=1)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv3x3_gn_fold_gelu_stats_residual(dt):
    """GroupNorm(1,C) folded into weights + border-class shift tables == conv(GN(x)); odd spatial size."""
    h = H()
    B, Cin, Cout, Hh, Ww = 3, 96, 192, 9, 7
    x = synth_input("k_f_x", (B, Cin, Hh, Ww)) * 2.0 + 0.7          # non-zero mean exercises the shift term
    w = synth_input("k_f_w", (Cout, Cin, 3, 3), 0.05)
    b = synth_input("k_f_b", (Cout,))
    g = 1 + 0.2 * synth_input("k_f_g", (Cin,))
    be = 0.3 * synth_input("k_f_be", (Cin,))
    r = synth_input("k_f_r", (B, Cout, Hh, Ww))
    want = F.gelu(F.conv2d(F.group_norm(x, 1, g, be, 1e-5), w, b, padding=1)) + r
    pc = h.PackedConv(w, b, dt, L.TILE_128x192, gamma=g, beta=be)
    y, st = h.run_conv(pc, h.to_nhwc(x, dt), pad=1, gn_ab=h.gn_ab_of(x), act=L.ACT_GELU, res=h.to_nhwc(r, dt), want_stats=True)
    assert rel_err(h.from_nhwc(y), want) < TOL[dt]
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 0], want.double().flatten(1).sum(1), rtol=5e-3 if dt else 1e-4, atol=1e-2)
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=5e-3 if dt else 1e-4)


@pytest.mark.parametrize("dt", DTS)
def test_conv1x1_two_source_concat_with_padding(dt):
    """res_conv over pad_and_concat(enc, dec) without materialising it (components:210-249)."""
    h = H()
    enc = synth_input("k_cc_e", (2, 96, 9, 7))
    dec = synth_input("k_cc_d", (2, 192, 8, 4))
    w = synth_input("k_cc_w", (96, 288, 1, 1), 0.1)
    b = synth_input("k_cc_b", (96,))
    dh, dw = 9 - 8, 7 - 4
    cat = torch.cat([enc, F.pad(dec, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))], 1)
    pc = h.PackedConv(w, b, dt, L.TILE_256x96)
    y, _ = h.run_conv(pc, h.to_nhwc(enc, dt), h.to_nhwc(dec, dt), off1=(dh // 2, dw // 2))
    assert rel_err(h.from_nhwc(y), F.conv2d(cat, w, b)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv_gn_fold_1x1(dt):
    h = H()
    x = synth_input("k_q_x", (2, 96, 8, 16)) + 0.5
    w = synth_input("k_q_w", (384, 96, 1, 1), 0.1)
    g = 1 + 0.2 * synth_input("k_q_g", (96,))
    be = 0.3 * synth_input("k_q_be", (96,))
    pc = h.PackedConv(w, None, dt, L.TILE_128x192, gamma=g, beta=be)
    y, _ = h.run_conv(pc, h.to_nhwc(x, dt), gn_ab=h.gn_ab_of(x))
    assert rel_err(h.from_nhwc(y), F.conv2d(F.group_norm(x, 1, g, be, 1e-5), w)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("hw", [(16, 10), (9, 7)])
def test_downsample_and_upsample(dt, hw):
    h = H()
    x = synth_input("k_du_x", (2, 96, *hw))
    w = synth_input("k_du_w", (96, 96, 4, 4), 0.05)
    b = synth_input("k_du_b", (96,))
    pc = h.PackedConv(w, b, dt, L.TILE_256x96)
    y, _ = h.run_conv(pc, h.to_nhwc(x, dt), stride=2, pad=1)
    assert rel_err(h.from_nhwc(y), F.conv2d(x, w, b, stride=2, padding=1)) < TOL[dt]
    pt = h.PackedConv(w, b, dt, L.TILE_256x96, transposed=True)
    y, _ = h.run_conv(pt, h.to_nhwc(x, dt))
    assert rel_err(h.from_nhwc(y), F.conv_transpose2d(x, w, b, stride=2, padding=1)) < TOL[dt]


@pytest.mark.parametrize("ks", [2, 3, 4])
def test_generic_conv_split_k(ks):
    """Split-K of the generic kernel (4x4 stride 2, transposed 4x4, folded 1x1): K slices + reduce == unsplit layer."""
    h = H()
    dt = L.DS_BF16
    x = synth_input("k_gs_x", (2, 192, 10, 14)) * 1.2 + 0.3
    w = synth_input("k_gs_w", (192, 192, 4, 4), 0.03)
    b = synth_input("k_gs_b", (192,))
    xd = h.to_nhwc(x, dt)
    xq = h.from_nhwc(xd)
    wq = w.bfloat16().float()
    pc = h.PackedConv(w, b, dt, L.TILE_64x192)
    y, st = h.run_conv(pc, xd, stride=2, pad=1, want_stats=True, ksplit=ks)
    want = F.conv2d(xq, wq, b, stride=2, padding=1)
    assert rel_err(h.from_nhwc(y), want) < TOL[dt]
    np.testing.assert_allclose(st.double().sum(1).cpu()[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)
    pt = h.PackedConv(w, b, dt, L.TILE_128x192, transposed=True)
    y, _ = h.run_conv(pt, xd, ksplit=ks)
    assert rel_err(h.from_nhwc(y), F.conv_transpose2d(xq, wq, b, stride=2, padding=1)) < TOL[dt]
    # 1x1 with the GroupNorm fold, residual and an odd pixel count
    w1 = synth_input("k_gs_w1", (96, 192, 1, 1), 0.1)
    b1 = synth_input("k_gs_b1", (96,))
    g = 1 + 0.2 * synth_input("k_gs_g", (192,))
    be = 0.3 * synth_input("k_gs_be", (192,))
    r = synth_input("k_gs_r", (2, 96, 10, 14))
    p1 = h.PackedConv(w1, b1, dt, L.TILE_256x96, gamma=g, beta=be)
    y, _ = h.run_conv(p1, xd, gn_ab=h.gn_ab_of(xq), res=h.to_nhwc(r, dt), ksplit=2)
    want = F.conv2d(F.group_norm(xq, 1, g, be, 1e-5), w1, b1) + h.from_nhwc(h.to_nhwc(r, dt))
    assert rel_err(h.from_nhwc(y), want) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_init_conv7x7_and_final_conv_nchw(dt):
    h = H()
    x = synth_input("k_i_x", (2, 4, 16, 12))
    w = synth_input("k_i_w", (96, 4, 7, 7), 0.1)
    b = synth_input("k_i_b", (96,))
    cp = 8 if dt == L.DS_BF16 else 4
    pc = h.PackedConv(w, b, dt, L.TILE_256x96, cin_pad=cp)
    y, _ = h.run_conv(pc, h.to_nhwc(x, dt, cp), pad=3)
    assert rel_err(h.from_nhwc(y), F.conv2d(x, w, b, padding=3)) < TOL[dt]
    x2 = synth_input("k_o_x", (2, 96, 16, 12))
    w2 = synth_input("k_o_w", (4, 96, 3, 3), 0.05)
    b2 = synth_input("k_o_b", (4,))
    pc2 = h.PackedConv(w2, b2, dt, L.TILE_128x32)
    y2, _ = h.run_conv(pc2, h.to_nhwc(x2, dt), pad=1, nchw_out=True)
    assert y2.dtype == torch.float32 and rel_err(y2.cpu(), F.conv2d(x2, w2, b2, padding=1)) < TOL[dt]


def test_conv_rejects_bad_arguments():
    h = H()
    w = synth_input("k_bad_w", (192, 64, 3, 3))
    pc = h.PackedConv(w, None, L.DS_F32, L.TILE_128x192)
    x = torch.zeros(1, 8, 8, 62, device="cuda")   # 62 channels: not a multiple of 4 and != packed Cin
    with pytest.raises(L.DsError):
        h.run_conv(pc, x, pad=1)


def test_conv_rejects_fields_its_tile_ignores():
    """flags (split precision) and a fused res_conv exist on the HALO3 / QUAD tiles only: any other tile must refuse them
    instead of computing a plain convolution."""
    import ctypes as C
    h = H()
    w = synth_input("k_bad2_w", (96, 96, 3, 3))
    x = torch.zeros(1, 8, 8, 96, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(1, 8, 8, 96, device="cuda", dtype=torch.bfloat16)
    aux = torch.zeros(96, device="cuda")

    def params(pc, **kw):
        p = L.ConvParams(src0=x.data_ptr(), src1=None, C0=96, C1=0, H=8, W=8, H1=0, W1=0, off_h1=0, off_w1=0, wpk=pc.w.data_ptr(),
                         Cout=96, cout_pad=pc.cout_pad, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, Ho=8, Wo=8, transposed=0,
                         out=out.data_ptr(), out_C=96, out_c0=0, out_nchw_f32=0, bias=None, gn_ab=None, fold_t1=None, fold_t2=None,
                         ncls=1, act=L.ACT_NONE, res=None, stats_part=None, B=1, dtype=L.DS_BF16, tile=pc.tile)
        p.wk_order = pc.k_order
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    for tile in (L.TILE_256x96, L.TILE_128x192):
        pc = h.PackedConv(w, None, L.DS_BF16, tile)
        L.call("ds_conv_igemm", C.byref(params(pc)), L.current_stream())                     # the plain launch is fine
        with pytest.raises(L.DsError, match="flags"):
            L.call("ds_conv_igemm", C.byref(params(pc, flags=4)), L.current_stream())
    for tile in (L.TILE_256x96, L.TILE_128x192):
        pc = h.PackedConv(w, None, L.DS_BF16, tile)
        with pytest.raises(L.DsError, match="res_conv"):
            L.call("ds_conv_igemm", C.byref(params(pc, res_steps=3, res_src0=x.data_ptr(), res_C0=96)), L.current_stream())
        with pytest.raises(L.DsError, match="res_steps = 0"):
            L.call("ds_conv_igemm", C.byref(params(pc, res_bias=aux.data_ptr())), L.current_stream())
    h.sync()


@pytest.mark.parametrize("case", ["qkv_fold", "res_two_source", "to_out_residual"])
def test_conv1x1_x3_split_precision(case):
    """ds_conv1x1_x3 (tier bf16x3): 1x1 convolution of fp32 tensors as x_hi w_hi + x_lo w_hi + x_hi w_lo on bf16 MFMAs vs torch fp64 —
    to_qkv with the PreNorm folded (components:263 + 142-152), res_conv over pad_and_concat(encoder, decoder) with a smaller, offset
    decoder map (components:128,210-249), to_out with bias and an fp32 residual; ragged pixel count, statistics partials."""
    import ctypes as C
    from diffusynth_amd.engine import pack_x3_1x1
    B, Hh, Ww = 2, 9, 31                       # 279 pixels: a full 256-pixel block and a ragged one
    g = torch.Generator().manual_seed({"qkv_fold": 1, "res_two_source": 2, "to_out_residual": 3}[case])
    rn = lambda *s: torch.randn(*s, generator=g)
    if case == "qkv_fold":
        C0, C1, Cout = 96, 0, 384
    elif case == "res_two_source":
        C0, C1, Cout = 96, 192, 96
    else:
        C0, C1, Cout = 128, 0, 192
    x0 = rn(B, C0, Hh, Ww) * 3.0
    w = rn(Cout, C0 + C1, 1, 1) * (C0 + C1) ** -0.5
    bias = rn(Cout)
    x1 = None
    off = (0, 0)
    xin = x0
    if C1:
        x1 = rn(B, C1, Hh - 1, Ww - 3)
        off = (0, 1)                          # pad_to_match: left = delta // 2
        pad = torch.zeros(B, C1, Hh, Ww)
        pad[:, :, off[0]:off[0] + Hh - 1, off[1]:off[1] + Ww - 3] = x1
        xin = torch.cat([x0, pad], 1)
    gamma = beta = ab = t1 = t2 = None
    if case == "qkv_fold":
        gamma, beta = rn(C0) * 0.5 + 1.0, rn(C0) * 0.3
        want = F.conv2d(F.group_norm(xin.double(), 1, gamma.double(), beta.double(), 1e-5), w.double())
    else:
        want = F.conv2d(xin.double(), w.double(), bias.double())
    res = None
    if case == "to_out_residual":
        res = rn(B, Cout, Hh, Ww)
        want = want + res.double()
    wpk, cout_pad = pack_x3_1x1(w.cuda(), gamma.cuda() if gamma is not None else None)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xd0, xd1 = nhwc(x0), (nhwc(x1) if x1 is not None else None)
    out = torch.full((B, Hh, Ww, Cout), float("nan"), device="cuda")
    p = L.ConvParams(src0=xd0.data_ptr(), src1=L.ptr(xd1), C0=C0, C1=C1, H=Hh, W=Ww, H1=(Hh - 1 if C1 else 0), W1=(Ww - 3 if C1 else 0),
                     off_h1=off[0], off_w1=off[1], wpk=wpk.data_ptr(), Cout=Cout, cout_pad=cout_pad, KH=1, KW=1, stride=1, pad_h=0, pad_w=0,
                     Ho=Hh, Wo=Ww, transposed=0, out=out.data_ptr(), out_C=Cout, out_c0=0, out_nchw_f32=0, bias=None, gn_ab=None,
                     fold_t1=None, fold_t2=None, ncls=1, act=L.ACT_NONE, res=None, stats_part=None, B=B, dtype=L.DS_BF16, tile=0)
    p.flags = 8 | 4
    if case == "qkv_fold":
        ab = H().gn_ab_of(xin)
        wd, gd, bd = w.float().contiguous().cuda(), gamma.cuda(), beta.cuda()
        t1, t2 = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
        L.call("ds_conv_fold_tables", wd.data_ptr(), None, gd.data_ptr(), bd.data_ptr(), Cout, C0, 1, 1, t1.data_ptr(), t2.data_ptr(), L.current_stream())
        p.gn_ab, p.fold_t1, p.fold_t2 = ab.data_ptr(), t1.data_ptr(), t2.data_ptr()
    else:
        bd = bias.cuda()
        p.bias = bd.data_ptr()
    if res is not None:
        rd = nhwc(res)
        p.res = rd.data_ptr()
    parts = L.load().ds_conv1x1_x3_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    L.call("ds_conv1x1_x3", C.byref(p), L.current_stream())
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    err = rel_err(got, want)
    print(f"conv1x1_x3 {case}: rel err {err:.2e}")
    assert err < 2e-5
    s = st.cpu().double().sum(1)
    assert torch.allclose(s[:, 0], want.sum((1, 2, 3)), rtol=1e-4, atol=1e-2) and torch.allclose(s[:, 1], (want ** 2).sum((1, 2, 3)), rtol=1e-4)
    # split-K (r04): K slices -> fp32 slab -> ds_conv_splitk_reduce (fold / bias, residual, statistics); ragged last slice included
    nq = (C0 + C1) // 32
    for ks in (2, 3, 4, 6, 8):
        if (ks - 1) * (-(-nq // ks)) >= nq:
            continue
        out2 = torch.full((B, Hh, Ww, Cout), float("nan"), device="cuda")
        slab = torch.full((ks, B, Hh, Ww, Cout), float("nan"), device="cuda")
        q2 = L.ConvParams.from_buffer_copy(p)
        q2.out, q2.ksplit, q2.slab, q2.stats_part = out2.data_ptr(), ks, slab.data_ptr(), None
        st2 = torch.zeros(B, L.load().ds_conv1x1_x3_stats_parts(C.byref(q2)), 2, device="cuda")
        q2.stats_part = st2.data_ptr()
        L.call("ds_conv1x1_x3", C.byref(q2), L.current_stream())
        L.call("ds_conv_splitk_reduce", C.byref(q2), L.current_stream())
        torch.cuda.synchronize()
        err2 = rel_err(out2.permute(0, 3, 1, 2).cpu(), want)
        assert torch.isfinite(slab).all() and err2 < 2e-5, (ks, err2)
        s2 = st2.cpu().double().sum(1)
        assert torch.allclose(s2[:, 1], (want ** 2).sum((1, 2, 3)), rtol=1e-4)
    bad = L.ConvParams.from_buffer_copy(p)
    bad.flags = 4
    with pytest.raises(L.DsError, match="fp32 in"):
        L.call("ds_conv1x1_x3", C.byref(bad), L.current_stream())


# ----------------------------------------------------------------------------------------- depthwise + GN
@pytest.mark.parametrize("dt", DTS)
def test_dwconv7_two_source_time_bias_stats(dt):
    h = H()
    B = 2
    enc = synth_input("k_dw_e", (B, 96, 10, 9))
    dec = synth_input("k_dw_d", (B, 192, 9, 6))
    w = synth_input("k_dw_w", (288, 1, 7, 7), 0.2)
    b = synth_input("k_dw_b", (288,))
    tb = synth_input("k_dw_tb", (B, 300))
    dh, dw = 1, 3
    cat = torch.cat([enc, F.pad(dec, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))], 1)
    want = F.conv2d(cat, w, b, padding=3, groups=288) + tb[:, 5:293, None, None]
    x0, x1 = h.to_nhwc(enc, dt), h.to_nhwc(dec, dt)
    wt = torch.empty(49 * 288, device="cuda")
    wd = w.contiguous().cuda()
    L.call("ds_pack_dw_weight", wd.data_ptr(), 288, wt.data_ptr(), L.current_stream())
    bd, tbd = b.cuda(), tb.cuda().contiguous()
    out = torch.empty(B, 10, 9, 288, device="cuda").to(h.TDT[dt])
    p = L.DwconvParams(src0=x0.data_ptr(), src1=x1.data_ptr(), C0=96, C1=192, H=10, W=9, H1=9, W1=6, off_h1=dh // 2,
                       off_w1=dw // 2, wt=wt.data_ptr(), bias=bd.data_ptr(), tbias=tbd.data_ptr() + 4 * 5, tb_stride=300,
                       out=out.data_ptr(), stats_part=None, B=B, dtype=dt)
    parts = L.load().ds_dwconv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    L.call("ds_dwconv7", C.byref(p), L.current_stream())
    h.sync()
    assert rel_err(h.from_nhwc(out), want) < (1e-5 if dt == L.DS_F32 else 1e-2)
    ab = torch.empty(B, 2, device="cuda")
    L.call("ds_gn_finalize", st.data_ptr(), B, parts, float(288 * 90), 1e-5, ab.data_ptr(), L.current_stream())
    h.sync()
    ref_ab = h.gn_ab_of(want).cpu()
    np.testing.assert_allclose(ab.cpu(), ref_ab, rtol=2e-3 if dt else 1e-5, atol=1e-4 if dt else 1e-6)


@pytest.mark.parametrize("hw,c01,split", [((64, 16), (96, 0), 1), ((70, 37), (32, 64), 1), ((48, 64), (64, 32), 1), ((33, 20), (96, 96), 0),
                                          ((256, 64), (96, 0), 1)])
def test_dwconv7_strip_kernel_matches_tile_kernel(hw, c01, split):
    """r05: the depthwise kernel that walks an LDS ring of 22 input rows down a 16-column strip (every input row leaves HBM once) against
    the tile kernel on the same operands: identical outputs bit for bit (same per-output operation order), statistics sums equal to fp32
    rounding, and both against torch's depthwise convolution.  Ragged heights / widths, two sources with pad offsets, image cut into row
    ranges (the 256-row case at B = 1), plane and fp32 outputs."""
    h = H()
    dt = L.DS_F32
    (Hh, Ww), (c0, c1) = hw, c01
    B, Cc = (1 if Hh >= 256 else 2), c0 + c1
    enc = synth_input("k_ds_e%s" % (hw,), (B, c0, Hh, Ww))
    w = synth_input("k_ds_w%d" % Cc, (Cc, 1, 7, 7), 0.2)
    b = synth_input("k_ds_b%d" % Cc, (Cc,))
    tb = synth_input("k_ds_tb", (B, Cc + 12))
    dh, dw = 1, 3
    x0 = h.to_nhwc(enc, dt)
    if c1:
        dec = synth_input("k_ds_d%s" % (hw,), (B, c1, Hh - 1, Ww - 3))
        x1 = h.to_nhwc(dec, dt)
        cat = torch.cat([enc, F.pad(dec, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))], 1)
    else:
        x1, cat = None, enc
    want = F.conv2d(cat.double(), w.double(), b.double(), padding=3, groups=Cc) + tb[:, 5:5 + Cc, None, None].double()
    wt = torch.empty(49 * Cc, device="cuda")
    wd = w.contiguous().cuda()
    L.call("ds_pack_dw_weight", wd.data_ptr(), Cc, wt.data_ptr(), L.current_stream())
    bd, tbd = b.cuda(), tb.cuda().contiguous()
    outs, sums = [], []
    for strip in (2, 1):                       # 2 = the tile kernel, 1 = the strip kernel
        out = torch.full((B, Hh, Ww, Cc), float("nan"), device="cuda")
        p = L.DwconvParams(src0=x0.data_ptr(), src1=(x1.data_ptr() if c1 else None), C0=c0, C1=c1, H=Hh, W=Ww, H1=(Hh - 1 if c1 else 0),
                           W1=(Ww - 3 if c1 else 0), off_h1=dh // 2, off_w1=dw // 2, wt=wt.data_ptr(), bias=bd.data_ptr(), tbias=tbd.data_ptr() + 4 * 5,
                           tb_stride=Cc + 12, out=out.data_ptr(), stats_part=None, B=B, dtype=dt, out_split=split, strip=strip)
        parts = L.load().ds_dwconv_stats_parts(C.byref(p))
        st = torch.zeros(B, parts, 2, device="cuda")
        p.stats_part = st.data_ptr()
        L.call("ds_dwconv7", C.byref(p), L.current_stream())
        h.sync()
        if split:
            pl = out.view(torch.bfloat16).view(B, Hh, Ww, 2 * Cc).float()
            val = pl[..., :Cc] + pl[..., Cc:]
        else:
            val = out
        assert torch.isfinite(val).all()
        outs.append(out.clone())
        sums.append((parts, st.double().sum(1).cpu()))
        assert rel_err(h.from_nhwc(val), want) < 1e-5, strip
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert sums[0][0] != sums[1][0]                                  # (a different number of partials: the strip kernel did run)
    np.testing.assert_allclose(sums[1][1], sums[0][1], rtol=1e-5)
    np.testing.assert_allclose(sums[1][1][:, 1], (want ** 2).flatten(1).sum(1), rtol=1e-4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("G,act", [(1, L.ACT_NONE), (8, L.ACT_SILU), (16, L.ACT_RELU)])
def test_gn_stats_and_apply(dt, G, act):
    h = H()
    B, Cc, Hh, Ww = 2, 160 if G == 16 else 96, 6, 10
    x = synth_input("k_gn_x", (B, Cc, Hh, Ww)) * 1.5 + 0.3
    r = synth_input("k_gn_r", (B, Cc, Hh, Ww))
    cb = synth_input("k_gn_cb", (B, Cc))
    g = 1 + 0.2 * synth_input("k_gn_g", (Cc,))
    be = 0.3 * synth_input("k_gn_b", (Cc,))
    eps = 1e-6 if G == 16 else 1e-5
    xd, rd = h.to_nhwc(x, dt), h.to_nhwc(r, dt)
    xq = h.from_nhwc(xd)          # statistics are taken over the stored (possibly bf16-rounded) values
    y = F.group_norm(xq, G, g, be, eps)
    y = {L.ACT_NONE: y, L.ACT_SILU: F.silu(y), L.ACT_RELU: F.relu(y)}[act]
    want = y + cb[:, :, None, None] + h.from_nhwc(rd)
    ab = torch.empty(B, G, 2, device="cuda")
    L.call("ds_gn_stats", xd.data_ptr(), dt, B, Hh * Ww, Cc, G, eps, ab.data_ptr(), L.current_stream())
    # the streaming form (what the engines launch) gives the same statistics
    ab2 = torch.empty(B, G, 2, device="cuda")
    ws = torch.empty(L.load().ds_gn_stats_ws_floats(B, Hh * Ww, Cc), device="cuda")
    L.call("ds_gn_stats_stream", xd.data_ptr(), dt, B, Hh * Ww, Cc, G, eps, ws.data_ptr(), ab2.data_ptr(), L.current_stream())
    h.sync()
    assert rel_err(ab2, ab) < 1e-5
    out = torch.empty_like(xd)
    gd, bd, cbd = g.cuda(), be.cuda(), cb.cuda().contiguous()
    p = L.GnApplyParams(x=xd.data_ptr(), res=rd.data_ptr(), out=out.data_ptr(), gn_ab=ab.data_ptr(), gamma=gd.data_ptr(),
                        beta=bd.data_ptr(), cbias=cbd.data_ptr(), cb_stride=Cc, B=B, HW=Hh * Ww, C=Cc, G=G, act=act, dtype=dt)
    L.call("ds_gn_apply", C.byref(p), L.current_stream())
    h.sync()
    assert rel_err(h.from_nhwc(out), want) < (1e-5 if dt == L.DS_F32 else 1e-2)


# ----------------------------------------------------------------------------------------- attention
def _attention(dt, qkv_nchw, heads, nseg, lq=None, lk=None, lv=None, q_softmax=1, scale=32 ** -0.5):
    h = H()
    B, _, Hh, Ww = qkv_nchw.shape
    N = Hh * Ww
    qd = h.to_nhwc(qkv_nchw, dt)
    part = torch.empty(L.load().ds_linattn_part_floats(B, heads, nseg), device="cuda")
    ctx = torch.empty(B, heads, 32, 32, device="cuda")
    out = torch.empty(B, Hh, Ww, heads * 32, device="cuda").to(h.TDT[dt])
    dev = lambda t: t.cuda().contiguous() if t is not None else None
    lq, lk, lv = dev(lq), dev(lk), dev(lv)
    p = L.AttnParams(qkv=qd.data_ptr(), B=B, N=N, heads=heads, dtype=dt, nseg=nseg, part=part.data_ptr(), ctx=ctx.data_ptr(),
                     label_q=L.ptr(lq), label_k=L.ptr(lk), label_v=L.ptr(lv), lq_stride=heads * 32, lk_stride=heads * 32,
                     lv_stride=heads * 32, q_softmax=q_softmax, scale=scale, out=out.data_ptr())
    L.call("ds_linattn_context", C.byref(p), L.current_stream())
    L.call("ds_linattn_output", C.byref(p), L.current_stream())
    h.sync()
    return h.from_nhwc(out), h.from_nhwc(qd)


def _attention_ref(qkv, heads, lq=None, lk=None, lv=None, q_softmax=True, scale=32 ** -0.5):
    b, _, hh, ww = qkv.shape
    q, k, v = (t.reshape(b, heads, 32, hh * ww) for t in qkv.chunk(3, dim=1))
    if lq is not None:
        q = q + lq.view(b, heads, 32, 1)
    if lk is not None:
        k = torch.cat([k, lk.view(b, heads, 32, 1)], -1)
        v = torch.cat([v, lv.view(b, heads, 32, 1)], -1)
    if q_softmax:
        q = q.softmax(dim=-2) * scale
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    return torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * 32, hh, ww)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", ["add", "add_nocond", "cat", "vqgan", "spike"])
def test_linear_attention(dt, case):
    heads = 1 if case == "vqgan" else 4
    B, Hh, Ww = 2, 24, 20                      # N = 480: ragged against the 64-pixel tile and the segments
    qkv = synth_input("k_at_qkv_" + case, (B, 3 * heads * 32, Hh, Ww)) * 2.0
    if case == "spike":                          # force the cross-segment max rescale (rule: test the rare branch)
        qkv[:, heads * 32 + 3, 5, 7] += 30.0
        qkv[:, heads * 32 + 40, 20, 1] += 45.0
    lq = synth_input("k_at_lq", (B, heads * 32)) if case in ("add", "spike") else None
    lk = synth_input("k_at_lk", (B, heads * 32)) * 3 if case == "cat" else None
    lv = synth_input("k_at_lv", (B, heads * 32)) if case == "cat" else None
    qs = case != "vqgan"
    for nseg in (1, 3):
        got, qkv_q = _attention(dt, qkv, heads, nseg, lq, lk, lv, int(qs), 32 ** -0.5 if qs else 1.0)
        want = _attention_ref(qkv_q, heads, lq, lk, lv, qs, 32 ** -0.5)
        assert rel_err(got, want) < (2e-5 if dt == L.DS_F32 else 1e-2), (case, nseg)


# ----------------------------------------------------------------------------------------- conditioning
def test_sinusoid_and_linear():
    from oracle.unet_ref import sinusoid
    t = torch.tensor([0, 1, 500, 999, 37], dtype=torch.long)
    half = 48
    import math
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    out = torch.empty(5, 96, device="cuda")
    td, fd = t.cuda(), freqs.cuda()
    L.call("ds_sinusoid", td.data_ptr(), fd.data_ptr(), 5, half, out.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    assert (out.cpu() - sinusoid(t, 96)).abs().max().item() < 2e-6
    x = synth_input("k_l_x", (5, 384))
    W = synth_input("k_l_w", (777, 384), 0.05)
    b = synth_input("k_l_b", (777,))
    y = torch.empty(5, 800, device="cuda")
    xd, Wd, bd = x.cuda(), W.cuda(), b.cuda()
    for act, fn in ((L.ACT_NONE, lambda v: v), (L.ACT_GELU, F.gelu), (L.ACT_SILU, F.silu)):
        L.call("ds_linear", xd.data_ptr(), 384, Wd.data_ptr(), bd.data_ptr(), 5, 384, 777, act, y.data_ptr(), 800, L.current_stream())
        torch.cuda.synchronize()
        assert rel_err(y[:, :777].cpu(), F.linear(fn(x), W, b)) < 1e-5
        # the hoisted form the engine uses for the time-bias stack: ds_activation once, then ds_linear without act_in — bit-identical
        if act != L.ACT_NONE:
            xa, y2 = torch.empty_like(xd), torch.empty_like(y)
            L.call("ds_activation", xd.data_ptr(), xd.numel(), act, xa.data_ptr(), L.current_stream())
            L.call("ds_linear", xa.data_ptr(), 384, Wd.data_ptr(), bd.data_ptr(), 5, 384, 777, L.ACT_NONE, y2.data_ptr(), 800, L.current_stream())
            torch.cuda.synchronize()
            assert torch.equal(y2[:, :777], y[:, :777])


def test_layout_roundtrip():
    x = synth_input("k_lay", (3, 4, 10, 6))
    xd = x.cuda()
    for dt, cp in ((L.DS_F32, 4), (L.DS_BF16, 8)):
        buf = torch.empty(3, 10, 6, cp, device="cuda").to(torch.float32 if dt == L.DS_F32 else torch.bfloat16)
        L.call("ds_nchw_to_nhwc", xd.data_ptr(), 3, 4, 10, 6, buf.data_ptr(), cp, dt, L.current_stream())
        back = torch.empty(3, 4, 10, 6, device="cuda")
        L.call("ds_nhwc_to_nchw", buf.data_ptr(), dt, 3, 4, cp, 10, 6, back.data_ptr(), L.current_stream())
        torch.cuda.synchronize()
        want = x if dt == L.DS_F32 else x.bfloat16().float()
        assert torch.equal(back.cpu(), want)
        assert torch.equal(buf[..., :4].float().permute(0, 3, 1, 2).cpu(), want)
        if cp > 4:
            assert buf[..., 4:].float().abs().max().item() == 0.0


# ----------------------------------------------------------------------------------------- sampler step
@pytest.mark.parametrize("cfg", [1.0, 6.0])
@pytest.mark.parametrize("blend", [0, 1, 2])
def test_ddim_step_bit_exact(cfg, blend):
    from oracle import sampler_ref as S
    B, Cc, Hh, Ww = 3, 4, 8, 12
    x, e_u, e_c, nz = (synth_input("k_s_" + n, (B, Cc, Hh, Ww)) for n in ("x", "eu", "ec", "nz"))
    guide, init = synth_input("k_s_g", (B, Cc, Hh, Ww)), synth_input("k_s_i", (B, Cc, Hh, Ww))
    mask = (synth_input("k_s_m", (B, 1, Hh, Ww)) > 0).float()
    s = S.RefSampler(1000)
    s.respace(list(np.linspace(0, 999, 20, dtype=np.int32)))
    for eta in (0.0, 1.0):
        for ti in (0, 1, 10, 19):
            t = torch.full((B,), ti, dtype=torch.long)
            eps = S.cfg_combine(e_u, e_c, cfg) if cfg != 1.0 else e_u
            want = S.ddim_update(x, eps, nz, s.alphas_cumprod, s.alphas_cumprod_prev, t, eta)
            tq = torch.clamp(t - 1, min=0)
            if blend == 1:
                want = mask * s.q_sample(guide, tq, noise=init) + (1 - mask) * want
            elif blend == 2:
                want = mask * guide + (1 - mask) * want
            a_t = torch.from_numpy(s.alphas_cumprod)[t].float()
            a_p = torch.from_numpy(s.alphas_cumprod_prev)[t].float()
            sig = eta * torch.sqrt((1 - a_p) / (1 - a_t)) * torch.sqrt(1 - a_t / a_p)
            coef = torch.stack([torch.sqrt(1. - a_t), torch.sqrt(a_t), torch.sqrt(a_p), torch.sqrt(1 - a_p - sig ** 2), sig], 1)
            qc = torch.stack([torch.from_numpy(s.sched["sqrt_alphas_cumprod"])[tq].float(),
                              torch.from_numpy(s.sched["sqrt_one_minus_alphas_cumprod"])[tq].float()], 1)
            d = {k: v.cuda().contiguous() for k, v in dict(x=x, eu=e_u, ec=e_c, nz=nz, g=guide, i=init, m=mask, c=coef, q=qc).items()}
            out = torch.empty_like(d["x"])
            p = L.StepParams(x=d["x"].data_ptr(), eps=d["eu"].data_ptr(), eps_cond=d["ec"].data_ptr() if cfg != 1.0 else None,
                             noise=d["nz"].data_ptr(), out=out.data_ptr(), coef=d["c"].data_ptr(), cfg_scale=cfg,
                             blend_mode=blend, guide=d["g"].data_ptr(), init_noise=d["i"].data_ptr(), mask=d["m"].data_ptr(),
                             qcoef=d["q"].data_ptr(), B=B, CHW=Cc * Hh * Ww, HW=Hh * Ww)
            L.call("ds_ddim_step", C.byref(p), L.current_stream())
            torch.cuda.synchronize()
            bad = (out.cpu() != want)
            assert not bad.any(), (cfg, blend, eta, ti, int(bad.sum()), (out.cpu() - want).abs().max().item())


def test_philox_normal_and_gather():
    n = 1 << 20
    a = torch.empty(n, device="cuda")
    b = torch.empty(n, device="cuda")
    L.call("ds_philox_normal", a.data_ptr(), n, 1234, 0, L.current_stream())
    L.call("ds_philox_normal", b.data_ptr(), n, 1234, 0, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a ** 4).mean().item() - 3.0) < 0.05                                # kurtosis of N(0,1)
    L.call("ds_philox_normal", b.data_ptr(), n // 2, 1234, n // 8, L.current_stream())   # offset continues the stream
    torch.cuda.synchronize()
    assert torch.equal(b[: n // 2], a[n // 2:])
    src = synth_input("k_gc", (6, 64)).cuda()
    cols = torch.tensor([0, 1, 2, 40, 41, 63, 5], dtype=torch.int32, device="cuda")
    out = torch.empty(6, 7, device="cuda")
    L.call("ds_gather_cols", src.data_ptr(), 6, 64, cols.data_ptr(), 7, out.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(out, src[:, cols.long()])


# ----------------------------------------------------------------------------------------- tail
def test_vq_nearest(vqgan_sd):
    from oracle import vqgan_ref as Q
    cb = vqgan_sd["_vq_vae._embedding.weight"]
    z = synth_input("k_vq_z", (3, 4, 16, 24)) * 1.3
    q_ref, _, _, idx_ref = Q.vq_forward(cb, z)
    zd, cbd = z.cuda(), cb.cuda().contiguous()
    esq = torch.sum(cb ** 2, dim=1).cuda()
    q = torch.empty_like(zd)
    idx = torch.empty(3 * 16 * 24, dtype=torch.int64, device="cuda")
    L.call("ds_vq_nearest", zd.data_ptr(), cbd.data_ptr(), esq.data_ptr(), 3, 4, 16 * 24, cb.shape[0], q.data_ptr(), idx.data_ptr(),
           L.current_stream())
    torch.cuda.synchronize()
    agree = (idx.cpu() == idx_ref).float().mean().item()
    assert agree > 0.999, agree
    # where the index differs the two codes must be equidistant to rounding (argmin tie, SURVEY "VQ argmin ties")
    flat = z.permute(0, 2, 3, 1).reshape(-1, 4).double()
    d_got = ((flat - cb[idx.cpu()].double()) ** 2).sum(1)
    d_ref = ((flat - cb[idx_ref].double()) ** 2).sum(1)
    assert ((d_got - d_ref).abs() <= 1e-5 * d_ref.clamp_min(1e-6)).all()
    same = (idx.cpu() == idx_ref).view(3, 16, 24)[:, None].expand_as(z)
    assert torch.equal(q.cpu()[same], q_ref[same])


def test_vq_nearest_first_minimum_on_exact_ties():
    """torch.argmin semantics of VectorQuantizer(EMA).forward (VQGAN.py:98-146): of several codes at exactly the same distance the FIRST wins —
    duplicated codebook rows that land in different code tiles and lane groups of the matrix-core search, and a ragged pixel count."""
    g = torch.Generator().manual_seed(9)
    cb = torch.randn(8192, 4, generator=g)
    dup = [(37, 5000), (1024, 1027), (15, 16), (4100, 8191)]
    for a, b in dup:
        cb[b] = cb[a]
    B, HW = 1, 77
    z = torch.randn(B, 4, HW, generator=g)
    for k, (a, _) in enumerate(dup):
        z[0, :, 5 + 16 * k] = cb[a]                        # exactly on a duplicated code: distance ties at the minimum
    zd, cbd = z.cuda().contiguous(), cb.cuda().contiguous()
    esq = torch.sum(cb ** 2, dim=1).cuda()
    q = torch.empty_like(zd)
    idx = torch.empty(B * HW, dtype=torch.int64, device="cuda")
    L.call("ds_vq_nearest", zd.data_ptr(), cbd.data_ptr(), esq.data_ptr(), B, 4, HW, 8192, q.data_ptr(), idx.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    for k, (a, _) in enumerate(dup):
        assert idx[5 + 16 * k].item() == a
    ref = ((z[0].t()[:, None, :].double() - cb[None].double()) ** 2).sum(-1).argmin(1)
    assert (idx.cpu() == ref).float().mean().item() > 0.98
    zt = z[0].t()
    assert torch.equal(q.cpu()[0].t(), zt + (cb[idx.cpu()] - zt))              # the straight-through form of VQGAN.py:140, bit for bit


def test_decoder_tail_and_istft():
    from oracle import vocoder_ref as V
    B, Fq, T = 2, 512, 12
    raw = synth_input("k_tail_raw", (B, 5, Fq, T)) * 2
    rd = raw.permute(0, 2, 3, 1).contiguous().cuda()
    enc = torch.empty(B, 3, Fq, T, device="cuda")
    L.call("ds_decoder_tail", rd.data_ptr(), L.DS_F32, B, 5, Fq * T, enc.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    want = torch.stack([F.softplus(raw[:, 0]), torch.tanh(raw[:, 1]), torch.tanh(raw[:, 2])], 1)
    assert rel_err(enc.cpu(), want) < 1e-6
    ws = torch.empty(L.load().ds_istft_ws_floats(B, Fq, T), device="cuda")
    audio = torch.empty(B, 256 * (T - 1), device="cuda")
    encd = want.cuda().contiguous()
    L.call("ds_istft_plus", encd.data_ptr(), B, Fq, T, 256, ws.data_ptr(), audio.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    ref = np.stack(V.latents_to_audio(want.numpy()))
    assert rel_err(audio.cpu(), ref) < 1e-4    # fp32 FFT on device vs the float64 CPU oracle


# ----------------------------------------------------------------------------------------- 3x3 halo kernel
@pytest.mark.parametrize("tile,cout", [(L.TILE_HALO3_256x96, 192), (L.TILE_HALO3_256x96, 96), (L.TILE_HALO3_256x96, 384), (L.TILE_HALO3_256x96, 64)])
@pytest.mark.parametrize("shape", [(2, 96, 8, 64), (2, 64, 16, 32), (1, 160, 37, 16), (3, 32, 33, 8), (1, 96, 9, 27), (1, 32, 5, 100), (2, 64, 7, 3)])
def test_conv3x3_halo_matches_conv2d(tile, cout, shape):
    """LDS-halo 3x3 kernel on every patch geometry (TW = 32/16/8), ragged H/W, W > 32 (several column tiles), Cout below a whole N-block."""
    h = H()
    dt = L.DS_BF16
    B, Cin, Hh, Ww = shape
    x = synth_input("k_h_x%s" % (shape,), shape) * 1.5 + 0.4
    w = synth_input("k_h_w%d_%d" % (cout, Cin), (cout, Cin, 3, 3), 0.05)
    b = synth_input("k_h_b%d" % cout, (cout,))
    g = 1 + 0.2 * synth_input("k_h_g%d" % Cin, (Cin,))
    be = 0.3 * synth_input("k_h_be%d" % Cin, (Cin,))
    r = synth_input("k_h_r%s%d" % (shape, cout), (B, cout, Hh, Ww))
    xd = h.to_nhwc(x, dt)
    xq = h.from_nhwc(xd)
    want = F.gelu(F.conv2d(F.group_norm(xq, 1, g, be, 1e-5), w, b, padding=1)) + h.from_nhwc(h.to_nhwc(r, dt))
    pc = h.PackedConv(w, b, dt, tile, gamma=g, beta=be)
    y, st = h.run_conv(pc, xd, pad=1, gn_ab=h.gn_ab_of(xq), act=L.ACT_GELU, res=h.to_nhwc(r, dt), want_stats=True)
    assert rel_err(h.from_nhwc(y), want) < TOL[dt]
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 0], want.double().flatten(1).sum(1), rtol=1e-2, atol=0.5)
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)
    # no activation + residual (conv2 of a ConvNeXt block): the line-sized epilogue (halo3_epilogue_rows: result staged as fp32 in LDS, residual
    # loaded, added and rounded on the contiguous side) — same reference, same statistics contract
    want_n = F.conv2d(F.group_norm(xq, 1, g, be, 1e-5), w, b, padding=1) + h.from_nhwc(h.to_nhwc(r, dt))
    yn, stn = h.run_conv(pc, xd, pad=1, gn_ab=h.gn_ab_of(xq), act=L.ACT_NONE, res=h.to_nhwc(r, dt), want_stats=True)
    assert rel_err(h.from_nhwc(yn), want_n) < TOL[dt]
    sn = stn.double().sum(1).cpu()
    np.testing.assert_allclose(sn[:, 0], want_n.double().flatten(1).sum(1), rtol=1e-2, atol=0.5)
    np.testing.assert_allclose(sn[:, 1], (want_n.double() ** 2).flatten(1).sum(1), rtol=1e-2)
    # the same weights through the generic im2col kernel agree to bf16 rounding (chunk-major packings are repacked tap-major)
    gen = L.TILE_128x192 if cout % 192 == 0 else L.TILE_256x96
    if pc.k_order:
        pc = h.PackedConv(w, b, dt, gen, gamma=g, beta=be)
    pc.tile = gen
    y2, _ = h.run_conv(pc, xd, pad=1, gn_ab=h.gn_ab_of(xq), act=L.ACT_GELU, res=h.to_nhwc(r, dt))
    assert rel_err(h.from_nhwc(y), h.from_nhwc(y2)) < 1e-2


@pytest.mark.parametrize("tile,cout,ks", [(L.TILE_HALO3_256x96, 384, 2), (L.TILE_HALO3_256x96, 192, 4), (L.TILE_HALO3_256x96, 96, 4), (L.TILE_HALO3_256x96, 64, 2),
                                          (L.TILE_HALO3_256x96, 192, 3), (L.TILE_HALO3_256x96, 96, 6)])
def test_conv3x3_halo_split_k(tile, cout, ks):
    """K split over blocks + reduce/epilogue kernel == unsplit result (to fp32 summation-order rounding before the bf16 store)."""
    h = H()
    dt = L.DS_BF16
    B, Cin, Hh, Ww = 2, 384, 20, 8
    x = synth_input("k_sk_x", (B, Cin, Hh, Ww)) * 1.5 + 0.4
    w = synth_input("k_sk_w%d" % cout, (cout, Cin, 3, 3), 0.05)
    b = synth_input("k_sk_b%d" % cout, (cout,))
    g = 1 + 0.2 * synth_input("k_sk_g", (Cin,))
    be = 0.3 * synth_input("k_sk_be", (Cin,))
    r = synth_input("k_sk_r%d" % cout, (B, cout, Hh, Ww))
    xd, rd = h.to_nhwc(x, dt), h.to_nhwc(r, dt)
    xq = h.from_nhwc(xd)
    want = F.gelu(F.conv2d(F.group_norm(xq, 1, g, be, 1e-5), w, b, padding=1)) + h.from_nhwc(rd)
    pc = h.PackedConv(w, b, dt, tile, gamma=g, beta=be)
    y, st = h.run_conv(pc, xd, pad=1, gn_ab=h.gn_ab_of(xq), act=L.ACT_GELU, res=rd, want_stats=True, ksplit=ks)
    assert rel_err(h.from_nhwc(y), want) < TOL[dt]
    y1, _ = h.run_conv(pc, xd, pad=1, gn_ab=h.gn_ab_of(xq), act=L.ACT_GELU, res=rd)
    assert rel_err(h.from_nhwc(y), h.from_nhwc(y1)) < 1e-2
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)


@pytest.mark.parametrize("mode,cin,cout,hw", [("up", 192, 96, (8, 32)), ("up", 192, 192, (9, 27)), ("up", 384, 96, (32, 8)), ("up", 192, 96, (5, 100)),
                                              ("down", 96, 96, (16, 64)), ("down", 96, 192, (18, 54)), ("down", 192, 96, (64, 16)), ("down", 96, 96, (10, 200))])
def test_conv_quad_halo3_matches_torch(mode, cin, cout, hw):
    """Conv2d(4, 2, 1) / ConvTranspose2d(4, 2, 1) on the four-tap halo kernel (DS_CONV_TILE_QUAD_HALO3) against torch on the
    bf16-rounded operands: every tile width (32 / 16 / 8), ragged grids, grids wider than one tile, all four phases / parity planes."""
    import ctypes as C
    from diffusynth_amd.engine import pack_quad_weights
    h = H()
    dt = L.DS_BF16
    B, (Hh, Ww) = 2, hw
    tr = mode == "up"
    x = synth_input("k_q_x%s%d" % (hw, cin), (B, cin, Hh, Ww)) * 1.5 + 0.3
    w = synth_input("k_q_w%s%d_%d" % (mode, cin, cout), (cin, cout, 4, 4) if tr else (cout, cin, 4, 4), 0.05)
    bb = synth_input("k_q_b%d" % cout, (cout,))
    xd = h.to_nhwc(x, dt)
    xq = h.from_nhwc(xd)
    wq = w.to(torch.bfloat16).float()
    want = F.conv_transpose2d(xq, wq, bb, stride=2, padding=1) if tr else F.conv2d(xq, wq, bb, stride=2, padding=1)
    wpk, cout_pad = pack_quad_weights(w.cuda(), tr)
    oh, ow = (2 * Hh, 2 * Ww) if tr else (Hh // 2, Ww // 2)
    gh, gw = (Hh, Ww) if tr else (oh, ow)
    out = torch.full((B, oh, ow, cout), float("nan"), device="cuda").to(h.TDT[dt])
    bd = bb.cuda()
    p = L.ConvParams(src0=xd.data_ptr(), src1=None, C0=cin, C1=0, H=Hh, W=Ww, H1=0, W1=0, off_h1=0, off_w1=0, wpk=wpk.data_ptr(), Cout=cout,
                     cout_pad=cout_pad, KH=2 if tr else 4, KW=2 if tr else 4, stride=1 if tr else 2, pad_h=0 if tr else 1, pad_w=0 if tr else 1,
                     Ho=gh, Wo=gw, transposed=1 if tr else 0, out=out.data_ptr(), out_C=cout, out_c0=0, out_nchw_f32=0, bias=bd.data_ptr(),
                     gn_ab=None, fold_t1=None, fold_t2=None, ncls=1, act=L.ACT_NONE, res=None, stats_part=None, B=B, dtype=dt,
                     tile=L.TILE_QUAD_HALO3, wk_order=2)
    lib = L.load()
    parts = lib.ds_conv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    L.call("ds_conv_igemm", C.byref(p), L.current_stream())
    torch.cuda.synchronize()
    got = h.from_nhwc(out)
    assert got.shape == want.shape
    assert rel_err(got, want) < TOL[dt]
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)
    # split-K (r04): K slices of whole groups of six chunks -> fp32 slab -> ds_conv_splitk_reduce (bias, bf16 store, statistics)
    nch = (1 if tr else 4) * cin // 32
    for ks in (2, 3, 4, 6, 8):
        if nch % ks or (nch // ks) % 6:
            continue
        out2 = torch.full((B, oh, ow, cout), float("nan"), device="cuda").to(h.TDT[dt])
        slab = torch.full((ks, B, oh, ow, (cout + 7) // 8 * 8), float("nan"), device="cuda")
        p.out, p.ksplit, p.slab, p.stats_part = out2.data_ptr(), ks, slab.data_ptr(), None
        parts2 = lib.ds_conv_stats_parts(C.byref(p))
        st2 = torch.zeros(B, parts2, 2, device="cuda")
        p.stats_part = st2.data_ptr()
        L.call("ds_conv_igemm", C.byref(p), L.current_stream())
        L.call("ds_conv_splitk_reduce", C.byref(p), L.current_stream())
        torch.cuda.synchronize()
        got2 = h.from_nhwc(out2)
        assert torch.isfinite(slab).all() and rel_err(got2, want) < TOL[dt], (ks, rel_err(got2, want))
        assert rel_err(got2, got) < 1e-2
        np.testing.assert_allclose(st2.double().sum(1).cpu()[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)


def test_conv_quad_halo3_split_precision(mode="up"):
    """Down / Upsample in the split-precision tier: fp32 input re-stored as hi / lo planes (ds_split_planes), weights [W_hi|W_hi|W_lo] as
    quad tiles, fp32 output — against torch in float64 on the unrounded operands."""
    import ctypes as C
    from diffusynth_amd.engine import pack_quad_weights
    h = H()
    for mode, cin, cout, (Hh, Ww) in (("up", 192, 96, (9, 27)), ("down", 96, 192, (18, 54)), ("up", 384, 96, (32, 8))):
        tr = mode == "up"
        B = 2
        x = synth_input("k_qs_x%s%d" % (mode, cin), (B, cin, Hh, Ww)) * 1.5 + 0.3
        w = synth_input("k_qs_w%s%d_%d" % (mode, cin, cout), (cin, cout, 4, 4) if tr else (cout, cin, 4, 4), 0.05)
        bb = synth_input("k_qs_b%d" % cout, (cout,))
        want = (F.conv_transpose2d(x.double(), w.double(), bb.double(), stride=2, padding=1) if tr
                else F.conv2d(x.double(), w.double(), bb.double(), stride=2, padding=1))
        hi = w.bfloat16().float()
        wpk, cout_pad = pack_quad_weights(torch.cat([hi, hi, w - hi], 0 if tr else 1).cuda(), tr)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        xs = torch.empty(B, Hh, Ww, 2 * cin, dtype=torch.bfloat16, device="cuda")
        L.call("ds_split_planes", xd.data_ptr(), xs.data_ptr(), B * Hh * Ww, cin, L.current_stream())
        oh, ow = (2 * Hh, 2 * Ww) if tr else (Hh // 2, Ww // 2)
        gh, gw = (Hh, Ww) if tr else (oh, ow)
        out = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
        bd = bb.cuda()
        p = L.ConvParams(src0=xs.data_ptr(), src1=None, C0=2 * cin, C1=0, H=Hh, W=Ww, H1=0, W1=0, off_h1=0, off_w1=0, wpk=wpk.data_ptr(), Cout=cout,
                         cout_pad=cout_pad, KH=2 if tr else 4, KW=2 if tr else 4, stride=1 if tr else 2, pad_h=0 if tr else 1, pad_w=0 if tr else 1,
                         Ho=gh, Wo=gw, transposed=1 if tr else 0, out=out.data_ptr(), out_C=cout, out_c0=0, out_nchw_f32=0, bias=bd.data_ptr(),
                         gn_ab=None, fold_t1=None, fold_t2=None, ncls=1, act=L.ACT_NONE, res=None, stats_part=None, B=B, dtype=L.DS_BF16,
                         tile=L.TILE_QUAD_HALO3, wk_order=2, flags=1 | 4)
        L.call("ds_conv_igemm", C.byref(p), L.current_stream())
        torch.cuda.synchronize()
        got = out.permute(0, 3, 1, 2).cpu()
        assert got.shape == want.shape
        assert rel_err(got, want.float()) < 3e-5, (mode, cin, cout)
        nch = (1 if tr else 4) * 3 * cin // 32
        for ks in (2, 3, 4, 6, 8):
            if nch % ks or (nch // ks) % 6:
                continue
            out2 = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
            slab = torch.full((ks, B, oh, ow, (cout + 7) // 8 * 8), float("nan"), device="cuda")
            p.out, p.ksplit, p.slab = out2.data_ptr(), ks, slab.data_ptr()
            L.call("ds_conv_igemm", C.byref(p), L.current_stream())
            L.call("ds_conv_splitk_reduce", C.byref(p), L.current_stream())
            torch.cuda.synchronize()
            assert rel_err(out2.permute(0, 3, 1, 2).cpu(), want.float()) < 3e-5, (mode, cin, cout, ks)


@pytest.mark.parametrize("out_mode", ["split", "f32", "f32+res"])
@pytest.mark.parametrize("shape,cout", [((2, 96, 8, 64), 192), ((1, 64, 37, 16), 96), ((2, 32, 33, 8), 96), ((1, 96, 9, 27), 96),
                                        # r05, two samples per block (images of at most 16 x 8, batch >= 2): an odd batch leaves the last block half empty
                                        ((3, 96, 16, 8), 192), ((2, 64, 12, 7), 96), ((5, 32, 16, 5), 96)])
def test_conv3x3_halo3_split_precision(shape, cout, out_mode):
    """Split-precision 3x3 (DS_CONV_F_*): fp32 tensors on bf16 matrix cores as x_hi w_hi + x_lo w_hi + x_hi w_lo.  Input as hi / lo
    bf16 planes, GroupNorm fold, exact GELU; output as hi / lo planes or as fp32 (+ fp32 residual).  Against F.conv2d in float64
    on the UNROUNDED fp32 operands: the tier's tolerance is 1e-3, the kernel is two orders below."""
    import ctypes as C
    from diffusynth_amd.engine import split3_weight, to_split_planes
    h = H()
    B, Cin, Hh, Ww = shape
    x = synth_input("k_sp_x%s" % (shape,), shape) * 1.5 + 0.4
    w = synth_input("k_sp_w%d_%d" % (cout, Cin), (cout, Cin, 3, 3), 0.05)
    bb = synth_input("k_sp_b%d" % cout, (cout,))
    gam = 1 + 0.2 * synth_input("k_sp_g%d" % Cin, (Cin,))
    bet = 0.3 * synth_input("k_sp_be%d" % Cin, (Cin,))
    r = synth_input("k_sp_r%s%d" % (shape, cout), (B, cout, Hh, Ww))
    gelu = out_mode == "split"
    xn = x.permute(0, 2, 3, 1).contiguous()
    xs = to_split_planes(xn).cuda()
    xq = (xs[..., :Cin].float() + xs[..., Cin:].float()).permute(0, 3, 1, 2).cpu()          # what the kernel sees (x to 2^-17)
    want = F.conv2d(F.group_norm(xq.double(), 1, gam.double(), bet.double(), 1e-5), w.double(), bb.double(), padding=1)
    if gelu:
        want = F.gelu(want)
    if out_mode == "f32+res":
        want = want + r.double()
    pc = h.PackedConv(split3_weight(w, gam), bb, L.DS_BF16, L.TILE_HALO3_256x96)             # per chunk [W_hi | W_lo | W_hi], gain already folded
    t1, t2 = torch.empty(9 * cout, device="cuda"), torch.empty(9 * cout, device="cuda")
    wd, gd, bd = w.cuda().contiguous(), gam.cuda(), bet.cuda()
    L.call("ds_conv_fold_tables", wd.data_ptr(), pc.bias.data_ptr(), gd.data_ptr(), bd.data_ptr(), cout, Cin, 3, 3, t1.data_ptr(), t2.data_ptr(),
           L.current_stream())
    ab = h.gn_ab_of(xq)
    flags = 1 | (2 if out_mode == "split" else 4)
    if out_mode == "split":
        out = torch.full((B, Hh, Ww, 2 * cout), float("nan"), device="cuda").bfloat16()
        out_C = 2 * cout
    else:
        out = torch.full((B, Hh, Ww, cout), float("nan"), device="cuda")
        out_C = cout
    rd = r.permute(0, 2, 3, 1).contiguous().cuda() if out_mode == "f32+res" else None
    p = L.ConvParams(src0=xs.data_ptr(), src1=None, C0=2 * Cin, C1=0, H=Hh, W=Ww, H1=0, W1=0, off_h1=0, off_w1=0, wpk=pc.w.data_ptr(), Cout=cout,
                     cout_pad=pc.cout_pad, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, Ho=Hh, Wo=Ww, transposed=0, out=out.data_ptr(), out_C=out_C,
                     out_c0=0, out_nchw_f32=0, bias=pc.bias.data_ptr(), gn_ab=ab.data_ptr(), fold_t1=t1.data_ptr(), fold_t2=t2.data_ptr(),
                     ncls=9, act=L.ACT_GELU if gelu else L.ACT_NONE, res=L.ptr(rd), stats_part=None, B=B, dtype=L.DS_BF16,
                     tile=L.TILE_HALO3_256x96, wk_order=1, flags=flags)
    # whole-K launch, then (r04) the same layer as K slices + ds_conv_splitk_reduce: what the engine runs at small batches
    for ks in (1, 2, 3, 4, 6):
        if ks > 1 and (Cin // 32) % ks != 0:          # K slices = whole source chunks
            continue
        out.fill_(float("nan"))
        p.ksplit, p.slab, p.stats_part = ks, None, None
        slab = None
        if ks > 1:
            slab = torch.empty(ks * B * Hh * Ww * ((cout + 7) // 8 * 8), device="cuda")
            p.slab = slab.data_ptr()
        parts = L.load().ds_conv_stats_parts(C.byref(p))
        st = torch.zeros(B, parts, 2, device="cuda")
        p.stats_part = st.data_ptr()
        L.call("ds_conv_igemm", C.byref(p), L.current_stream())
        if ks > 1:
            L.call("ds_conv_splitk_reduce", C.byref(p), L.current_stream())
        torch.cuda.synchronize()
        got = out.float()
        if out_mode == "split":
            got = got[..., :cout] + got[..., cout:]
        got = got.permute(0, 3, 1, 2).cpu()
        assert rel_err(got, want.float()) < 3e-5, ks
        s = st.double().sum(1).cpu()
        np.testing.assert_allclose(s[:, 1], (want ** 2).flatten(1).sum(1), rtol=1e-4)


@pytest.mark.parametrize("shape,cout", [((2, 96, 8, 64), 4), ((1, 64, 37, 16), 16), ((2, 128, 33, 8), 3), ((1, 96, 9, 27), 4), ((1, 32, 5, 100), 8)])
def test_conv3x3_smalln_matches_conv2d(shape, cout):
    """Few-output 3x3 (DS_CONV_TILE_HALO3_N16, the final 96 -> 4 convolution): all chunk halos staged at once, weights in registers;
    1 / 2 / 3 / 4 chunks (4 = two groups), every tile width, ragged tiles, pad channels of the output written as zeros."""
    import ctypes as C
    h = H()
    dt = L.DS_BF16
    B, Cin, Hh, Ww = shape
    x = synth_input("k_sn_x%s" % (shape,), shape) * 1.5 + 0.4
    w = synth_input("k_sn_w%d_%d" % (cout, Cin), (cout, Cin, 3, 3), 0.05)
    bb = synth_input("k_sn_b%d" % cout, (cout,))
    xd = h.to_nhwc(x, dt)
    want = F.conv2d(h.from_nhwc(xd), w.to(torch.bfloat16).float(), bb, padding=1)
    lib = L.load()
    n = lib.ds_pack_conv_elems(Cin, 3, 3, 16, 0)
    wpk = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    wd, bd = w.cuda().contiguous(), bb.cuda()
    pp = L.PackConvParams(w=wd.data_ptr(), gamma=None, dst=wpk.data_ptr(), dtype=dt, Cout=cout, Cin=Cin, cin_pad=Cin, KH=3, KW=3, cout_pad=16,
                          transposed=0, k_order=1)
    L.call("ds_pack_conv_weight", C.byref(pp), L.current_stream())
    out_C = (cout + 7) // 8 * 8
    out = torch.full((B, Hh, Ww, out_C), float("nan"), device="cuda").to(torch.bfloat16)
    p = L.ConvParams(src0=xd.data_ptr(), src1=None, C0=Cin, C1=0, H=Hh, W=Ww, H1=0, W1=0, off_h1=0, off_w1=0, wpk=wpk.data_ptr(), Cout=cout,
                     cout_pad=16, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, Ho=Hh, Wo=Ww, transposed=0, out=out.data_ptr(), out_C=out_C, out_c0=0,
                     out_nchw_f32=0, bias=bd.data_ptr(), gn_ab=None, fold_t1=None, fold_t2=None, ncls=1, act=L.ACT_NONE, res=None,
                     stats_part=None, B=B, dtype=dt, tile=L.TILE_HALO3_N16, wk_order=1)
    L.call("ds_conv_igemm", C.byref(p), L.current_stream())
    torch.cuda.synchronize()
    got = h.from_nhwc(out)
    assert torch.isfinite(got).all()
    assert got[:, cout:].abs().max().item() == 0.0 if out_C > cout else True
    assert rel_err(got[:, :cout], want) < TOL[dt]


# ----------------------------------------------------------------------------------------- 7x7 init convolution
@pytest.mark.parametrize("hw,cin,cx", [((16, 32), 4, 8), ((37, 70), 4, 8), ((9, 5), 3, 4), ((256, 64), 4, 8)])
def test_conv7x7_c4_matches_torch(hw, cin, cx):
    """ds_conv7x7_c4 (the U-Net's init_conv on its own kernel) == F.conv2d(x, w, b, padding=3) on the bf16-rounded operands: ragged tiles,
    image borders (zeros from the buffer range check), a block that walks several tiles (batch 3 at 256x64 = 1536 tiles on 512 blocks)."""
    h = H()
    B, (Hh, Ww) = 3, hw
    x = synth_input("k_i7_x%s" % (hw,), (B, cin, Hh, Ww))
    w = synth_input("k_i7_w%d" % cin, (96, cin, 7, 7), 0.1)
    b = synth_input("k_i7_b", (96,))
    xp = torch.zeros(B, cx, Hh, Ww)
    xp[:, :cin] = x
    xp[:, cin:] = 7.0                                   # stored channels beyond the real ones: not read (>= 4) or met by zero weights
    xd = h.to_nhwc(xp, L.DS_BF16)
    want = F.conv2d(h.from_nhwc(xd)[:, :cin], w.bfloat16().float(), b, padding=3)
    wd, bd = w.contiguous().cuda(), b.cuda()
    wp = torch.empty(L.load().ds_conv7x7_c4_weight_elems(), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_conv7x7_c4", wd.data_ptr(), 96, cin, wp.data_ptr(), st)
    out = torch.full((B, Hh, Ww, 96), float("nan"), device="cuda").to(torch.bfloat16)
    L.call("ds_conv7x7_c4", xd.data_ptr(), B, Hh, Ww, cx, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st)
    h.sync()
    got = h.from_nhwc(out)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 6e-3, rel_err(got, want)


@pytest.mark.parametrize("hw,cin", [((256, 64), 4), ((37, 45), 4), ((9, 100), 3)])
def test_conv7x7_c4_x3_matches_torch(hw, cin):
    """ds_conv7x7_c4_x3 (the init convolution of the bf16x3 tier: fp32 in, fp32 out, x_hi w_hi + x_lo w_hi + x_hi w_lo on bf16 MFMAs) against
    F.conv2d in float64 on the UNROUNDED operands: the tier's tolerance is 1e-3, the kernel is two orders below."""
    B, (Hh, Ww) = 3, hw
    x = synth_input("k_i7x_x%s" % (hw,), (B, cin, Hh, Ww)) * 2.0 + 0.3
    w = synth_input("k_i7x_w%d" % cin, (96, cin, 7, 7), 0.1)
    b = synth_input("k_i7x_b", (96,))
    want = F.conv2d(x.double(), w.double(), b.double(), padding=3)
    xp = torch.zeros(B, 4, Hh, Ww)
    xp[:, :cin] = x
    xp[:, cin:] = 7.0                                   # a stored channel beyond the real ones meets zero weights
    xd = xp.permute(0, 2, 3, 1).contiguous().cuda()
    wd, bd = w.contiguous().cuda(), b.cuda()
    wp = torch.empty(2 * L.load().ds_conv7x7_c4_weight_elems(), dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_conv7x7_c4_x3", wd.data_ptr(), 96, cin, wp.data_ptr(), st)
    out = torch.full((B, Hh, Ww, 96), float("nan"), device="cuda")
    L.call("ds_conv7x7_c4_x3", xd.data_ptr(), B, Hh, Ww, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_err(got, want.float()) < 3e-5, rel_err(got, want.float())


# ----------------------------------------------------------------------------------------- fused attention block
@pytest.mark.parametrize("Cc,hw,cond", [(96, (16, 16), True), (96, (5, 10), False), (192, (33, 32), True), (384, (8, 6), True),
                                        (96, (193, 257), True), (192, (257, 259), False), (384, (32, 33), True)])
def test_fused_attention_block_matches_oracle(Cc, hw, cond):
    """ds_attn_fused_context/_output + gn_finalize + gn_apply == Residual(PreNorm(LinearCrossAttentionAdd)) of the oracle
    (bf16 tier; N ragged against the 32-pixel tiles and the segments).  The two large cases give every wave of the second-generation
    kernels several tiles at batch 2 (their grid is sized for the whole chip), which is what the headline batch runs; the 384-channel
    case at 1056 pixels takes the second-generation context pass with two heads per block."""
    from oracle import unet_ref as U
    from diffusynth_amd.synth import synth_state_dict
    h = H()
    B, (Hh, Ww) = 2, hw
    N = Hh * Ww
    tag = "fa%d" % Cc
    spec = [(tag + ".fn.fn.to_qkv.weight", (384, Cc, 1, 1)), (tag + ".fn.fn.to_out.0.weight", (Cc, 128, 1, 1)),
            (tag + ".fn.fn.to_out.0.bias", (Cc,)), (tag + ".fn.fn.to_out.1.weight", (Cc,)), (tag + ".fn.fn.to_out.1.bias", (Cc,)),
            (tag + ".fn.fn.label_key.weight", (128, 512)), (tag + ".fn.fn.label_key.bias", (128,)),
            (tag + ".fn.fn.label_query.weight", (128, 512)), (tag + ".fn.fn.label_query.bias", (128,)),
            (tag + ".fn.norm.weight", (Cc,)), (tag + ".fn.norm.bias", (Cc,))]
    sd = synth_state_dict(spec)
    x = synth_input("fa_x%d%s" % (Cc, hw), (B, Cc, Hh, Ww)) * 1.3 + 0.2
    c = synth_input("fa_c", (B, 512)) if cond else None
    xd = h.to_nhwc(x, L.DS_BF16)
    xq = h.from_nhwc(xd)
    want = U.attn_block(sd, tag, xq, c, "linear_add")
    dev = lambda t: t.float().contiguous().cuda()
    wq, wo = dev(sd[tag + ".fn.fn.to_qkv.weight"].reshape(384, Cc)), dev(sd[tag + ".fn.fn.to_out.0.weight"].reshape(Cc, 128))
    g, be = dev(sd[tag + ".fn.norm.weight"]), dev(sd[tag + ".fn.norm.bias"])
    wq16 = torch.empty(384 * Cc, dtype=torch.bfloat16, device="cuda")
    wo16 = torch.empty(Cc * 128, dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    L.call("ds_pack_attn_fused", wq.data_ptr(), g.data_ptr(), wo.data_ptr(), wq16.data_ptr(), wo16.data_ptr(), Cc, st)
    t1, t2 = torch.empty(384, device="cuda"), torch.empty(384, device="cuda")
    L.call("ds_conv_fold_tables", wq.data_ptr(), None, g.data_ptr(), be.data_ptr(), 384, Cc, 1, 1, t1.data_ptr(), t2.data_ptr(), st)
    lq = None
    if cond:
        lq = dev(F.linear(c, sd[tag + ".fn.fn.label_query.weight"], sd[tag + ".fn.fn.label_query.bias"]))
    for nseg, v2 in ((1, False), (3, False), (3, True)):       # v2: second-generation output pass (to_out folded into the context)
        ab = h.gn_ab_of(xq)
        part = torch.empty(L.load().ds_linattn_part_floats(B, 4, nseg), device="cuda")
        ctx = torch.empty(B * 4 * 1024, device="cuda")
        y = torch.empty(B, Hh, Ww, Cc, dtype=torch.bfloat16, device="cuda")
        bo = dev(sd[tag + ".fn.fn.to_out.0.bias"])
        p = L.AttnFusedParams(x=xd.data_ptr(), B=B, N=N, C=Cc, nseg=nseg, wqkv=wq16.data_ptr(), t1=t1.data_ptr(), t2=t2.data_ptr(),
                              gn_ab=ab.data_ptr(), label_q=L.ptr(lq), lq_stride=128, scale=32 ** -0.5, part=part.data_ptr(),
                              ctx=ctx.data_ptr(), wout_perm=wo16.data_ptr(), bias_out=bo.data_ptr(), y=y.data_ptr(), stats_part=None)
        mf = torch.empty(B * Cc * 128, dtype=torch.bfloat16, device="cuda") if v2 else None
        p.mfold = L.ptr(mf)
        p.gen = 2 if v2 else 1                          # (gen = 0 would pick the first generation at this batch)
        parts = L.load().ds_attn_fused_stats_parts(C.byref(p))
        sp = torch.zeros(B, parts, 2, device="cuda")
        p.stats_part = sp.data_ptr()
        L.call("ds_attn_fused_context", C.byref(p), st)
        L.call("ds_attn_fused_output", C.byref(p), st)
        aby = torch.empty(B, 2, device="cuda")
        L.call("ds_gn_finalize", sp.data_ptr(), B, parts, float(Cc * N), 1e-5, aby.data_ptr(), st)
        out = torch.empty_like(y)
        go, bo2 = dev(sd[tag + ".fn.fn.to_out.1.weight"]), dev(sd[tag + ".fn.fn.to_out.1.bias"])
        gp = L.GnApplyParams(x=y.data_ptr(), res=xd.data_ptr(), out=out.data_ptr(), gn_ab=aby.data_ptr(), gamma=go.data_ptr(),
                             beta=bo2.data_ptr(), cbias=None, cb_stride=0, B=B, HW=N, C=Cc, G=1, act=L.ACT_NONE, dtype=L.DS_BF16)
        L.call("ds_gn_apply", C.byref(gp), st)
        h.sync()
        err = rel_err(h.from_nhwc(out), want)
        assert err < 2e-2, (Cc, hw, cond, nseg, v2, err)


@pytest.mark.parametrize("Cc,hw,cond", [(96, (16, 16), True), (96, (5, 10), False), (192, (33, 32), True), (384, (8, 6), True),
                                        (96, (193, 257), True), (192, (129, 131), False), (384, (32, 33), True)])
def test_attn_x3_block_matches_oracle(Cc, hw, cond):
    """Split-precision fused attention (tier bf16x3, csrc/attn_x3.hip): ds_attn_x3_context/_output + ds_gn_apply on fp32 tensors ==
    Residual(PreNorm(LinearCrossAttentionAdd)) of the oracle (components:142-152,252-293) to 2e-5 — every product is three bf16 MFMA
    terms with fp32 accumulation.  N ragged against the 32-pixel tiles and the segments; the large cases give every wave several tiles;
    the GroupNorm partials of y must equal the statistics of the y that was written."""
    from oracle import unet_ref as U
    from diffusynth_amd.synth import synth_state_dict
    h = H()
    B, (Hh, Ww) = 2, hw
    N = Hh * Ww
    tag = "fx%d" % Cc
    spec = [(tag + ".fn.fn.to_qkv.weight", (384, Cc, 1, 1)), (tag + ".fn.fn.to_out.0.weight", (Cc, 128, 1, 1)),
            (tag + ".fn.fn.to_out.0.bias", (Cc,)), (tag + ".fn.fn.to_out.1.weight", (Cc,)), (tag + ".fn.fn.to_out.1.bias", (Cc,)),
            (tag + ".fn.fn.label_key.weight", (128, 512)), (tag + ".fn.fn.label_key.bias", (128,)),
            (tag + ".fn.fn.label_query.weight", (128, 512)), (tag + ".fn.fn.label_query.bias", (128,)),
            (tag + ".fn.norm.weight", (Cc,)), (tag + ".fn.norm.bias", (Cc,))]
    sd = synth_state_dict(spec)
    x = synth_input("fx_x%d%s" % (Cc, hw), (B, Cc, Hh, Ww)) * 1.3 + 0.2
    c = synth_input("fx_c", (B, 512)) if cond else None
    xd = h.to_nhwc(x, L.DS_F32)
    want = U.attn_block(sd, tag, x, c, "linear_add")
    dev = lambda t: t.float().contiguous().cuda()
    wq, wo = dev(sd[tag + ".fn.fn.to_qkv.weight"].reshape(384, Cc)), dev(sd[tag + ".fn.fn.to_out.0.weight"].reshape(Cc, 128))
    g, be = dev(sd[tag + ".fn.norm.weight"]), dev(sd[tag + ".fn.norm.bias"])
    whl = torch.empty(2 * 384 * Cc, dtype=torch.bfloat16, device="cuda")
    st = L.current_stream()
    lib = L.load()
    L.call("ds_pack_attn_x3", wq.data_ptr(), g.data_ptr(), whl.data_ptr(), Cc, st)
    t1, t2 = torch.empty(384, device="cuda"), torch.empty(384, device="cuda")
    L.call("ds_conv_fold_tables", wq.data_ptr(), None, g.data_ptr(), be.data_ptr(), 384, Cc, 1, 1, t1.data_ptr(), t2.data_ptr(), st)
    lq = dev(F.linear(c, sd[tag + ".fn.fn.label_query.weight"], sd[tag + ".fn.fn.label_query.bias"])) if cond else None
    bo = dev(sd[tag + ".fn.fn.to_out.0.bias"])
    go, bo2 = dev(sd[tag + ".fn.fn.to_out.1.weight"]), dev(sd[tag + ".fn.fn.to_out.1.bias"])
    for nseg in (1, 3, lib.ds_attn_x3_segments(B, N, Cc)):
        ab = h.gn_ab_of(x)
        part = torch.empty(lib.ds_linattn_part_floats(B, 4, nseg), device="cuda")
        ctx = torch.empty(B * 4 * 1024, device="cuda")
        qpl = torch.empty(lib.ds_attn_x3_qplane_bytes(B, N), dtype=torch.uint8, device="cuda")
        mf = torch.empty(lib.ds_attn_x3_mfold_bytes(B, Cc), dtype=torch.uint8, device="cuda")
        y = torch.full((B, Hh, Ww, Cc), float("nan"), device="cuda")
        p = L.AttnX3Params(x=xd.data_ptr(), B=B, N=N, C=Cc, nseg=nseg, wqkv_hl=whl.data_ptr(), t1=t1.data_ptr(), t2=t2.data_ptr(),
                           gn_ab=ab.data_ptr(), label_q=L.ptr(lq), lq_stride=128, scale=32 ** -0.5, part=part.data_ptr(),
                           ctx=ctx.data_ptr(), qplanes=qpl.data_ptr(), mfold=mf.data_ptr(), wout=wo.data_ptr(), bias_out=bo.data_ptr(),
                           y=y.data_ptr(), stats_part=None)
        parts = lib.ds_attn_x3_stats_parts(C.byref(p))
        sp = torch.zeros(B, parts, 2, device="cuda")
        p.stats_part = sp.data_ptr()
        L.call("ds_attn_x3_context", C.byref(p), st)
        L.call("ds_attn_x3_output", C.byref(p), st)
        aby = torch.empty(B, 2, device="cuda")
        L.call("ds_gn_finalize", sp.data_ptr(), B, parts, float(Cc * N), 1e-5, aby.data_ptr(), st)
        out = torch.empty_like(y)
        gp = L.GnApplyParams(x=y.data_ptr(), res=xd.data_ptr(), out=out.data_ptr(), gn_ab=aby.data_ptr(), gamma=go.data_ptr(),
                             beta=bo2.data_ptr(), cbias=None, cb_stride=0, B=B, HW=N, C=Cc, G=1, act=L.ACT_NONE, dtype=L.DS_F32)
        L.call("ds_gn_apply", C.byref(gp), st)
        h.sync()
        assert torch.isfinite(y).all()
        s = sp.double().sum(1).cpu()
        yd = y.double().cpu().reshape(B, -1)
        assert torch.allclose(s[:, 0], yd.sum(1), rtol=1e-5, atol=1e-2) and torch.allclose(s[:, 1], (yd * yd).sum(1), rtol=1e-5)
        err = rel_err(h.from_nhwc(out), want)
        assert err < 2e-5, (Cc, hw, cond, nseg, err)
        # form B: the same block with the output GroupNorm + residual applied inside the second of two output passes (no y tensor)
        out_b = torch.full((B, Hh, Ww, Cc), float("nan"), device="cuda")
        sp.zero_()
        p.y, p.out, p.on_gamma, p.on_beta, p.on_eps = None, out_b.data_ptr(), go.data_ptr(), bo2.data_ptr(), 1e-5
        L.call("ds_attn_x3_context", C.byref(p), st)
        L.call("ds_attn_x3_output", C.byref(p), st)
        h.sync()
        err_b = rel_err(h.from_nhwc(out_b), want)
        assert err_b < 2e-5 and rel_err(out_b, out) < 2e-6, (Cc, hw, cond, nseg, err_b)
        # ... with the result also / only as hi / lo bf16 planes [B][N][2C] (what the Down / Upsample behind the block reads): hi + lo = out to 2^-17
        for keep_f32 in (True, False):
            out_c = torch.full((B, Hh, Ww, Cc), float("nan"), device="cuda")
            pl = torch.full((B, Hh, Ww, 2 * Cc), float("nan"), device="cuda").to(torch.bfloat16)
            sp.zero_()
            p.out, p.out_planes = (out_c.data_ptr() if keep_f32 else None), pl.data_ptr()
            L.call("ds_attn_x3_context", C.byref(p), st)
            L.call("ds_attn_x3_output", C.byref(p), st)
            h.sync()
            if keep_f32:
                assert torch.equal(out_c, out_b)
            hi, lo = pl[..., :Cc].float(), pl[..., Cc:].float()
            assert torch.equal(hi, out_b.bfloat16().float()) and rel_err(hi + lo, out_b) < 1e-5, (Cc, hw, keep_f32, rel_err(hi + lo, out_b))
        p.out_planes = None


@pytest.mark.parametrize("hw", [(10, 9), (16, 32), (37, 70), (40, 16), (33, 13)])
def test_dwconv7_mfma_two_source(hw):
    """Toeplitz/MFMA form of the depthwise 7x7 (bf16): two-source concat with padding offsets, ragged tiles, stats."""
    h = H()
    dt = L.DS_BF16
    B, (Hh, Ww) = 2, hw
    enc = synth_input("k_dm_e%s" % (hw,), (B, 96, Hh, Ww))
    dec = synth_input("k_dm_d%s" % (hw,), (B, 192, Hh - 1, Ww - 3))
    w = synth_input("k_dm_w", (288, 1, 7, 7), 0.2)
    b = synth_input("k_dm_b", (288,))
    tb = synth_input("k_dm_tb", (B, 300))
    dh, dw = 1, 3
    x0, x1 = h.to_nhwc(enc, dt), h.to_nhwc(dec, dt)
    cat = torch.cat([h.from_nhwc(x0), F.pad(h.from_nhwc(x1), (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))], 1)
    wq = w.bfloat16().float()                      # the MFMA path holds the taps in bf16
    want = F.conv2d(cat, wq, b, padding=3, groups=288) + tb[:, 5:293, None, None]
    wd = w.contiguous().cuda()
    wt = torch.empty(49 * 288, device="cuda")
    we = torch.empty(288 * 6 * 64 * 8, dtype=torch.bfloat16, device="cuda")
    L.call("ds_pack_dw_weight", wd.data_ptr(), 288, wt.data_ptr(), L.current_stream())
    L.call("ds_pack_dw_weight_mfma", wd.data_ptr(), 288, we.data_ptr(), L.current_stream())
    bd, tbd = b.cuda(), tb.cuda().contiguous()
    out = torch.full((B, Hh, Ww, 288), float("nan"), device="cuda").to(torch.bfloat16)
    p = L.DwconvParams(src0=x0.data_ptr(), src1=x1.data_ptr(), C0=96, C1=192, H=Hh, W=Ww, H1=Hh - 1, W1=Ww - 3, off_h1=dh // 2,
                       off_w1=dw // 2, wt=wt.data_ptr(), bias=bd.data_ptr(), tbias=tbd.data_ptr() + 4 * 5, tb_stride=300,
                       out=out.data_ptr(), stats_part=None, B=B, dtype=dt, wexp=we.data_ptr())
    parts = L.load().ds_dwconv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    L.call("ds_dwconv7", C.byref(p), L.current_stream())
    h.sync()
    assert rel_err(h.from_nhwc(out), want) < 1e-2
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=2e-2)


@pytest.mark.parametrize("tile", [L.TILE_HALO3_256x96])
@pytest.mark.parametrize("shape,cx", [((2, 192, 16, 32), (96, 0)), ((1, 64, 37, 16), (64, 32)), ((2, 96, 9, 27), (96, 96)), ((1, 32, 33, 8), (32, 64))])
def test_conv3x3_halo2_with_fused_res_conv(shape, cx, tile):
    """ConvNeXt conv2 + the block's 1x1 res_conv in ONE launch (components:125-139): 3x3 over GroupNorm(g) [folded], scaled in
    registers, then 1x1 over pad_and_concat(x0, x1) accumulated at the centre tap; x1 smaller than the image (pad offsets)."""
    import ctypes as C
    h = H()
    dt = L.DS_BF16
    B, Cin, Hh, Ww = shape
    cout, (c0, c1) = 96, cx
    g_in = synth_input("k_rf_g%s" % (shape,), shape) * 1.5 + 0.4
    w = synth_input("k_rf_w%d" % Cin, (cout, Cin, 3, 3), 0.05)
    bb = synth_input("k_rf_b", (cout,))
    gam = 1 + 0.2 * synth_input("k_rf_gam%d" % Cin, (Cin,))
    bet = 0.3 * synth_input("k_rf_bet%d" % Cin, (Cin,))
    wr = synth_input("k_rf_wr%d" % (c0 + c1), (cout, c0 + c1, 1, 1), 0.1)
    br = synth_input("k_rf_br", (cout,))
    x0 = synth_input("k_rf_x0%s" % (shape,), (B, c0, Hh, Ww))
    h1, w1, oh, ow = Hh - 2, Ww - 1, 1, 0                       # decoder map one pixel short: pad_and_concat offsets (1, 0)
    x1 = synth_input("k_rf_x1%s" % (shape,), (B, c1, h1, w1)) if c1 else None
    gd, x0d = h.to_nhwc(g_in, dt), h.to_nhwc(x0, dt)
    x1d = h.to_nhwc(x1, dt) if c1 else None
    gq, x0q = h.from_nhwc(gd), h.from_nhwc(x0d)
    xcat = x0q
    if c1:
        x1p = F.pad(h.from_nhwc(x1d), (ow, Ww - w1 - ow, oh, Hh - h1 - oh))
        xcat = torch.cat([x0q, x1p], 1)
    want = F.conv2d(F.group_norm(gq, 1, gam, bet, 1e-5), w, bb, padding=1) + F.conv2d(xcat, wr, br)
    pc = h.PackedConv(w, bb, dt, tile, gamma=gam, beta=bet)
    # the 1x1 tiles ([cx/32][cout_pad][32], k_order 1) go in front of the 3x3 tiles
    lib = L.load()
    n = lib.ds_pack_conv_elems(c0 + c1, 1, 1, pc.cout_pad, 0)
    rpk = torch.empty(n, dtype=h.TDT[dt], device="cuda")
    wrd = wr.float().contiguous().cuda()
    pp = L.PackConvParams(w=wrd.data_ptr(), gamma=None, dst=rpk.data_ptr(), dtype=dt, Cout=cout, Cin=c0 + c1, cin_pad=c0 + c1, KH=1, KW=1,
                          cout_pad=pc.cout_pad, transposed=0, k_order=1)
    L.call("ds_pack_conv_weight", C.byref(pp), L.current_stream())
    wall = torch.cat([rpk, pc.w])
    brd = br.cuda()
    ab = h.gn_ab_of(gq)
    out = torch.full((B, Hh, Ww, cout), float("nan"), device="cuda").to(h.TDT[dt])
    p = L.ConvParams(src0=gd.data_ptr(), src1=None, C0=Cin, C1=0, H=Hh, W=Ww, H1=0, W1=0, off_h1=0, off_w1=0, wpk=wall.data_ptr(), Cout=cout,
                     cout_pad=pc.cout_pad, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, Ho=Hh, Wo=Ww, transposed=0, out=out.data_ptr(), out_C=cout,
                     out_c0=0, out_nchw_f32=0, bias=pc.bias.data_ptr(), gn_ab=ab.data_ptr(), fold_t1=pc.t1.data_ptr(), fold_t2=pc.t2.data_ptr(),
                     ncls=9, act=L.ACT_NONE, res=None, stats_part=None, B=B, dtype=dt, tile=tile, wk_order=1,
                     res_src0=x0d.data_ptr(), res_src1=L.ptr(x1d), res_C0=c0, res_C1=c1, res_H1=h1 if c1 else 0, res_W1=w1 if c1 else 0,
                     res_off_h1=oh if c1 else 0, res_off_w1=ow if c1 else 0, res_steps=(c0 + c1) // 32, res_bias=brd.data_ptr())
    parts = lib.ds_conv_stats_parts(C.byref(p))
    st = torch.zeros(B, parts, 2, device="cuda")
    p.stats_part = st.data_ptr()
    L.call("ds_conv_igemm", C.byref(p), L.current_stream())
    torch.cuda.synchronize()
    assert rel_err(h.from_nhwc(out), want) < TOL[dt]
    s = st.double().sum(1).cpu()
    np.testing.assert_allclose(s[:, 1], (want.double() ** 2).flatten(1).sum(1), rtol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout", [((2, 96, 37, 19), 4), ((1, 32, 16, 16), 3), ((2, 64, 5, 70), 1)])
def test_conv3x3_f32_n4_matches_torch(shape, cout):
    """ds_conv3x3_f32_n4 (the fp32 tiers' final 96 -> 4 convolution: diffusion.py:103-105) against F.conv2d in float64: ragged tiles, one tile
    column / row, fewer than four outputs (the missing ones must be zero), fp32 products and sums (error at the 1e-6 level)."""
    B, Cc, Hh, Ww = shape
    x = synth_input("k_n4_x%s" % (shape,), shape) * 1.5 + 0.2
    w = synth_input("k_n4_w%d_%d" % (cout, Cc), (cout, Cc, 3, 3), 0.05)
    b = synth_input("k_n4_b%d" % cout, (cout,))
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd, bd = w.cuda().contiguous(), b.cuda()
    wpk = torch.empty(L.load().ds_conv3x3_f32_n4_weight_floats(Cc), device="cuda")
    out = torch.full((B, Hh, Ww, 4), float("nan"), device="cuda")
    L.call("ds_pack_conv3x3_f32_n4", wd.data_ptr(), bd.data_ptr(), cout, Cc, wpk.data_ptr(), L.current_stream())
    L.call("ds_conv3x3_f32_n4", xd.data_ptr(), B, Hh, Ww, Cc, wpk.data_ptr(), out.data_ptr(), L.current_stream())
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert rel_err(got[:, :cout], want.float()) < 5e-6
    assert (got[:, cout:] == 0).all()
    with pytest.raises(L.DsError, match="multiple of 32"):
        L.call("ds_conv3x3_f32_n4", xd.data_ptr(), B, Hh, Ww, 40, wpk.data_ptr(), out.data_ptr(), L.current_stream())
