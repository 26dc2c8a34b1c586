// 3x3 stride-1 pad-1 convolution with at most 16 output channels (the U-Net's final_conv: Conv2d(96, 4, 3, padding=1),
// diffusion.py final_conv[1]) — bf16, 16x16x32 MFMAs, HBM-bound by construction (gfx950).
//
// With N = 16 a K step is 4 MFMAs per wave: the ring / barrier-per-step pipeline of conv3x3_halo3.hip would spend its time in barriers
// (and the generic implicit-GEMM kernel ran this layer at 1 TB/s: 0.39 ms for 403 MB).  Here a block stages the input halo of up to
// three 32-channel chunks (= all 96 input channels of the layer) in LDS AT ONCE — the swizzled 64-byte-row image of the halo3 kernels,
// one image per chunk — and every wave keeps all 27 weight fragments (9 taps x 3 chunks, 16 rows x 32 k) in registers: after ONE
// barrier the block runs 108 ds_read_b128 + 108 MFMAs per wave with no synchronisation, then stores 8 bytes per pixel and lane group.
// Larger Cin loops over groups of three chunks.  Weights: chunk-major tiles [cc*9 + tap][cout_pad = 16][32] (wk_order = 1).
#include "common.hpp"
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif

#include "conv_halo3_common.hpp"

namespace {

constexpr int SN_CH = 3;                                   // chunks staged per group
constexpr int SN_HALO = 408 * PSTR;                        // one chunk image (34 x 12 pixels for the 8-wide tile)
constexpr int SN_LDS = SN_CH * SN_HALO;                    // 78336 <= 81920: two blocks per CU

template <int TWL>
__global__ __launch_bounds__(NT, DS_MINBLK) void conv3x3_smalln_kernel(const ds_conv_params p) {
    using G = HG<TWL>;
    constexpr int TW = G::TW, TH = G::TH, HCP = G::HCP, NPX = G::NPX, H_IT = G::H_IT;
    static_assert(NPX * PSTR <= SN_HALO, "chunk image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int tiles_w = (p.W + TW - 1) >> TWL;
    const int gx = gridDim.x, nwg = gx * gridDim.z;
    int wid = blockIdx.x + gx * blockIdx.z;
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);      // XCD-chunked order: neighbouring tiles on one L2
    const int bx = wid % gx, b = wid / gx;
    const int th = bx / tiles_w, tw = bx - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const int Cin = p.C0, NCC = Cin >> 5;

    const char* const hbase = reinterpret_cast<const char*>(p.src0) + (size_t)b * p.H * p.W * Cin * 2;
    const rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(hbase), (short)0, (int)((unsigned)p.H * p.W * Cin * 2), 0x00020000);
    const char* const wbase = reinterpret_cast<const char*>(p.wpk);
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), (short)0, (int)((unsigned)NCC * 9 * 16 * 64), 0x00020000);

    unsigned hvo[H_IT];
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
        const int slot = tid + it * NT, hp = slot >> 2, dq = slot & 3;
        const int hr = hp / HCP, hc = hp - hr * HCP;
        hvo[it] = VOFF_NONE;
        if (hp < NPX && hc < TW + 2) {
            const int hi = h0 + hr - 1, wi = w0 + hc - 1;
            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) hvo[it] = (unsigned)((hi * p.W + wi) * Cin + dq * 8) * 2u;
        }
    }
    const int lds_h = (tid >> 2) * PSTR + (((tid & 3) ^ (((tid >> 4) & 1) << 1)) << 4);
    int xb[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
        else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }
        const int hp0 = row_l * HCP + col_l;
        xb[i] = hp0 * PSTR + ((q ^ (((hp0 >> 2) & 1) << 1)) << 4);
    }
    const int xm1 = ((m & 3) == 3) << 5, xm2 = ((m & 3) >= 2) << 5;     // swizzle flips of the tap shifts (conv3x3_halo3.hip)

    f32x4 acc[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < NCC; g0 += SN_CH) {
        const int ng = NCC - g0 < SN_CH ? NCC - g0 : SN_CH;             // chunks of this group (block-uniform)
        if (g0 > 0) __syncthreads();                                     // the previous group's images have been read
        // ---- all loads of the group in one round trip: halo pieces, then this wave's weight fragments straight into registers
        u32x4 hreg[SN_CH][H_IT];
#pragma unroll
        for (int c = 0; c < SN_CH; ++c)
#pragma unroll
            for (int it = 0; it < H_IT; ++it) hreg[c][it] = buf_ld16(rs_h, hbase, c < ng ? hvo[it] : VOFF_NONE, (unsigned)(g0 + c) * 64u, DS_BX_SRC0);
        bf16x8 wf[SN_CH][9];
#pragma unroll
        for (int c = 0; c < SN_CH; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wf[c][t] = __builtin_bit_cast(bf16x8, buf_ld16(rs_w, wbase, c < ng ? (unsigned)(m * 64 + q * 16) : VOFF_NONE, (unsigned)((g0 + c) * 9 + t) * 1024u, DS_BX_W));
#pragma unroll
        for (int c = 0; c < SN_CH; ++c)
#pragma unroll
            for (int it = 0; it < H_IT; ++it)
                if (it * 64 * PSTR + 64 * PSTR <= SN_HALO || (tid >> 2) + it * 64 < SN_HALO / PSTR)
                    *reinterpret_cast<u32x4*>(smem + c * SN_HALO + lds_h + it * 64 * PSTR) = hreg[c][it];
        __syncthreads();
        // ---- 9 taps x ng chunks: 4 fragment reads + 4 MFMAs each, no synchronisation
#pragma unroll
        for (int c = 0; c < SN_CH; ++c) {
            if (c < ng) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ty = t / 3, tx = t % 3;
#pragma unroll
                    for (int i = 0; i < XT; ++i) {
                        int a = xb[i];
                        if (tx == 0 && ty == 1) a ^= 32;
                        if (tx == 1) a ^= (ty == 1 ? (xm1 ^ 32) : xm1);
                        if (tx == 2) a ^= (ty == 1 ? (xm2 ^ 32) : xm2);
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(smem + c * SN_HALO + a + (ty * HCP + tx) * PSTR);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][t], xf, acc[i], 0, 0, 0);      // D^T = W . X^T
                    }
                }
            }
        }
    }
    // ---- epilogue: lane = pixel (lane & 15) of each pixel tile, channels 4 * (lane >> 4) .. + 3: bias, bf16, 8-byte store
    const int c0 = 4 * q;
    const int cout_v = (p.Cout + 7) / 8 * 8;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c0 + r < p.Cout) bias[r] = DS_LD(float, p.bias + c0 + r, DS_BX_BIAS);
    }
    bf16* const outp = reinterpret_cast<bf16*>(p.out) + (size_t)b * p.H * p.W * p.out_C + p.out_c0 + c0;
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
        else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }
        const int ho = h0 + row_l, wo = w0 + col_l;
        if (ho < p.H && wo < p.W && c0 < cout_v) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][r] + bias[r];
                if (p.act == DS_ACT_GELU) v = gelu_fast(v);
                o[r] = (bf16)v;
            }
            DS_ST(bf16x4, outp + (size_t)(ho * p.W + wo) * p.out_C, DS_BX_OUT, o);
        }
    }
}

int smalln_twl(int W) {
    int twl = 3;
    while ((1 << twl) < W && twl < 5) ++twl;
    return twl;
}

}  // namespace

int ds_conv3x3_smalln_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv3x3_smalln: bf16 only");
    DS_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad_h == 1 && p->pad_w == 1 && !p->transposed && p->Ho == p->H && p->Wo == p->W,
               "conv3x3_smalln: 3x3 stride 1 pad 1 only");
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0 && p->Cout <= 16 && p->cout_pad == 16 && p->wk_order == 1,
               "conv3x3_smalln: single source, Cin %% 32 == 0, Cout <= 16 packed chunk-major with cout_pad = 16");
    DS_REQUIRE(!p->gn_ab && !p->gn_part && !p->res && !p->stats_part && p->ksplit <= 1 && !p->res_steps && !p->flags && !p->out_nchw_f32,
               "conv3x3_smalln: bias (+ GELU) epilogue only: no GroupNorm fold, residual, statistics or split-K");
    DS_REQUIRE((long long)p->H * p->W * p->C0 * 2 < (1ll << 31), "conv3x3_smalln: one input sample must stay below 2 GiB (32-bit buffer offsets)");
    const int twl = smalln_twl(p->W), TW = 1 << twl, TH = BM >> twl;
    dim3 grid(((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW), 1, p->B);
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_HALO);
        ds_conv_bounds_table(*p, DS_K_CONV_HALO, grid.x, &h.t);
        h.set(DS_BX_W, p->wpk, (long long)(p->C0 / 32) * 9 * 16 * 64);
        h.publish(st);
    }
#endif
    if (twl == 5) {
        DS_SET_MAX_LDS(conv3x3_smalln_kernel<5>, SN_LDS, "conv3x3_smalln<32>");
        hipLaunchKernelGGL(conv3x3_smalln_kernel<5>, grid, dim3(NT), SN_LDS, st, *p);
    } else if (twl == 4) {
        DS_SET_MAX_LDS(conv3x3_smalln_kernel<4>, SN_LDS, "conv3x3_smalln<16>");
        hipLaunchKernelGGL(conv3x3_smalln_kernel<4>, grid, dim3(NT), SN_LDS, st, *p);
    } else {
        DS_SET_MAX_LDS(conv3x3_smalln_kernel<3>, SN_LDS, "conv3x3_smalln<8>");
        hipLaunchKernelGGL(conv3x3_smalln_kernel<3>, grid, dim3(NT), SN_LDS, st, *p);
    }
    DS_CHECK_LAUNCH("conv3x3_smalln");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_smalln(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
