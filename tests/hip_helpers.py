"""Test-side wrappers that drive single kernels of libdiffusynth_hip.so through the C ABI.
Layout conversions here use torch (test plumbing only)."""
import ctypes as C

import torch

from diffusynth_amd import _lib as L

DEV = "cuda"
TDT = {L.DS_F32: torch.float32, L.DS_BF16: torch.bfloat16}


def sync():
    torch.cuda.synchronize()


def to_nhwc(x, dt, c_pad=None):
    """NCHW fp32 (cpu) -> NHWC device tensor of the kernel dtype, channels zero-padded to c_pad."""
    x = x.float()
    if c_pad is not None and c_pad > x.shape[1]:
        x = torch.cat([x, torch.zeros(x.shape[0], c_pad - x.shape[1], *x.shape[2:])], 1)
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(TDT[dt])


def from_nhwc(y):
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


def up(x, m):
    return (x + m - 1) // m * m


class PackedConv:
    def __init__(self, weight, bias, dt, tile, cin_pad=None, gamma=None, beta=None, transposed=False):
        lib = L.load()
        w = weight.float().contiguous().to(DEV)
        if transposed:
            self.Cin, self.Cout = w.shape[0], w.shape[1]
            self.KH = self.KW = 2
        else:
            self.Cout, self.Cin, self.KH, self.KW = w.shape
        self.cin_pad = self.Cin if cin_pad is None else cin_pad
        self.tile, self.dt, self.transposed = tile, dt, transposed
        self.k_order = 1 if tile == L.TILE_HALO3_256x96 else 0        # chunk-major K order for the hand-scheduled halo kernel
        bn = lib.ds_conv_tile_bn(tile)
        self.cout_pad = up(self.Cout, bn)
        n = lib.ds_pack_conv_elems(self.cin_pad, self.KH, self.KW, self.cout_pad, int(transposed))
        self.w = torch.empty(n, dtype=TDT[dt], device=DEV)
        g = gamma.float().contiguous().to(DEV) if gamma is not None else None
        pp = L.PackConvParams(w=w.data_ptr(), gamma=L.ptr(g), dst=self.w.data_ptr(), dtype=dt, Cout=self.Cout, Cin=self.Cin,
                              cin_pad=self.cin_pad, KH=self.KH, KW=self.KW, cout_pad=self.cout_pad, transposed=int(transposed),
                              k_order=self.k_order)
        L.call("ds_pack_conv_weight", C.byref(pp), L.current_stream())
        self.bias = bias.float().contiguous().to(DEV) if bias is not None else None
        self.t1 = self.t2 = None
        self.ncls = 1
        if gamma is not None:
            self.ncls = 9 if self.KH == 3 else 1
            self.t1 = torch.empty(self.ncls * self.Cout, device=DEV)
            self.t2 = torch.empty(self.ncls * self.Cout, device=DEV)
            b = beta.float().contiguous().to(DEV)
            L.call("ds_conv_fold_tables", w.data_ptr(), L.ptr(self.bias), g.data_ptr(), b.data_ptr(), self.Cout, self.Cin,
                   self.KH, self.KW, self.t1.data_ptr(), self.t2.data_ptr(), L.current_stream())
        sync()


def run_conv(pc, x0, x1=None, off1=(0, 0), stride=1, pad=0, gn_ab=None, act=L.ACT_NONE, res=None, want_stats=False,
             nchw_out=False, ksplit=1):
    """x0/x1/res: NHWC device tensors.  Returns (out NHWC or NCHW fp32, stats partial tensor or None)."""
    B, H, W, C0 = x0.shape
    C1 = x1.shape[3] if x1 is not None else 0
    if pc.transposed:
        Ho, Wo, oh, ow = H, W, 2 * H, 2 * W
    else:
        Ho = (H + 2 * pad - pc.KH) // stride + 1
        Wo = (W + 2 * pad - pc.KW) // stride + 1
        oh, ow = Ho, Wo
    vec = 8 if pc.dt == L.DS_BF16 else 4
    out_C = up(pc.Cout, vec)
    out = torch.full((B, oh, ow, out_C), float("nan"), device=DEV).to(TDT[pc.dt])
    p = L.ConvParams(src0=x0.data_ptr(), src1=L.ptr(x1), C0=C0, C1=C1, H=H, W=W,
                     H1=(x1.shape[1] if x1 is not None else 0), W1=(x1.shape[2] if x1 is not None else 0),
                     off_h1=off1[0], off_w1=off1[1], wpk=pc.w.data_ptr(), Cout=pc.Cout, cout_pad=pc.cout_pad, KH=pc.KH,
                     KW=pc.KW, stride=stride, pad_h=pad, pad_w=pad, Ho=Ho, Wo=Wo, transposed=int(pc.transposed),
                     out=out.data_ptr(), out_C=out_C, out_c0=0, out_nchw_f32=0, bias=L.ptr(pc.bias),
                     gn_ab=L.ptr(gn_ab), fold_t1=L.ptr(pc.t1) if gn_ab is not None else None,
                     fold_t2=L.ptr(pc.t2) if gn_ab is not None else None, ncls=pc.ncls if gn_ab is not None else 1,
                     act=act, res=L.ptr(res), stats_part=None, B=B, dtype=pc.dt, tile=pc.tile)
    p.wk_order = pc.k_order
    st = None
    slab = None
    if ksplit > 1:
        slab = torch.full((ksplit, B, oh * ow, up(pc.Cout, 8)), float("nan"), device=DEV)
        p.ksplit, p.slab = ksplit, slab.data_ptr()
    if want_stats:
        parts = L.load().ds_conv_stats_parts(C.byref(p))
        st = torch.zeros(B, parts, 2, device=DEV)
        p.stats_part = st.data_ptr()
    L.call("ds_conv_igemm", C.byref(p), L.current_stream())
    if ksplit > 1:
        L.call("ds_conv_splitk_reduce", C.byref(p), L.current_stream())
    sync()
    if nchw_out:       # the fp32 NCHW boundary is a separate converter kernel
        o2 = torch.empty(B, pc.Cout, oh, ow, device=DEV)
        L.call("ds_nhwc_to_nchw", out.data_ptr(), pc.dt, B, pc.Cout, out_C, oh, ow, o2.data_ptr(), L.current_stream())
        sync()
        return o2, st
    if out_C != pc.Cout:
        assert out[..., pc.Cout:].float().abs().max().item() == 0.0     # pad channels are exact zeros
        out = out[..., :pc.Cout].contiguous()
    return out, st


def gn_ab_of(x_nchw, eps=1e-5):
    """(rstd, rstd*mean) per sample of an NCHW cpu tensor, float64 math -> device fp32 [B][2]."""
    xd = x_nchw.double().flatten(1)
    mean = xd.mean(1)
    var = xd.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    return torch.stack([rstd, rstd * mean], 1).float().to(DEV).contiguous()
