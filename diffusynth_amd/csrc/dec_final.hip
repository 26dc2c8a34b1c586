// Last block of the VQGAN decoder in ONE kernel (bf16 activations; gfx950).
//
// Decoder.forward ends with ResnetBlock(80 -> 3) and the output activations (VQGAN.py:177-244, 390-398):
//     y = conv3x3(swish(GroupNorm16(x))) + nin_shortcut_1x1(x);   out = [softplus(y0), tanh(y1), tanh(y2)]
// on the largest tensor of the tail (512 x 256 x 80 per clip: 1.3 GB bf16 at batch 64).  As separate launches this stage read that tensor
// three times and wrote it once more (GroupNorm apply 0.42 ms, the two 80 -> 3 convolutions on the generic small-N tile 0.9 ms each, the
// activation kernel): 2.3 of the tail's 9.3 ms.  Here the tensor is read ONCE: a block stages the 34 x 10 halo of its 32 x 8 pixel tile for
// all 80 channels (three 32-channel chunk images in LDS, the swizzled 64-byte rows of conv3x3_halo3; channels 80..95 zero), applying the
// per-(sample, group) GroupNorm affine and the swish on the way; the 1x1 shortcut on the RAW values is taken in the same pass (each thread
// dots its 24 channels of a pixel, four adjacent lanes hold a pixel: two DPP adds) and parked in a 4 KB LDS buffer; the 3x3 runs as in
// conv3x3_smalln (27 weight fragments in registers, 108 ds_read_b128 + 108 MFMAs per wave, one barrier); the epilogue adds the two
// branches and the biases, applies softplus / tanh and stores fp32 NCHW planes.
#include "common.hpp"
#include "conv_halo3_common.hpp"

namespace {

constexpr int DF_CH = 3;                                    // 32-channel chunk images (Cin <= 96)

template <int TWL>
struct DF {
    using G = HG<TWL>;
    static constexpr int IMG = G::H_IT * 64 * PSTR;         // one chunk image: whole store iterations of 64 pixels
    static constexpr int OFF_TAB = DF_CH * IMG;             // [96] (scale, shift) fp32
    static constexpr int OFF_NIN = OFF_TAB + 96 * 8;        // [256 pixels][4] fp32: the shortcut branch
    static constexpr int LDS = OFF_NIN + BM * 16;
    static_assert(LDS <= 81920, "two blocks per CU");
};

struct ds_dec_final_params {
    const void* x; int B, H, W, C, G;
    const float* gn_ab; const float* gamma; const float* beta;
    const void* w3; const float* b3; const float* wnin; const float* bnin;
    float* out;
};

__device__ __forceinline__ float swish_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v)); }

template <int TWL>
__global__ __launch_bounds__(NT, DS_MINBLK) void dec_final_kernel(const ds_dec_final_params p) {
    using G = HG<TWL>;
    using D = DF<TWL>;
    constexpr int TW = G::TW, TH = G::TH, HCP = G::HCP, NPX = G::NPX, H_IT = G::H_IT, IMG = D::IMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + D::OFF_TAB);
    float* const nin = reinterpret_cast<float*>(smem + D::OFF_NIN);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int tiles_w = (p.W + TW - 1) >> TWL;
    const int gx = gridDim.x, nwg = gx * gridDim.z;
    int wid = blockIdx.x + gx * blockIdx.z;
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);      // XCD-chunked order: neighbouring tiles on one L2
    const int bx = wid % gx, b = wid / gx;
    const int th = bx / tiles_w, tw = bx - th * tiles_w, h0 = th * TH, w0 = tw * TW;
    const int C = p.C, HWp = p.H * p.W, npc = C >> 3;                   // 16-byte pieces per pixel
    const bf16* const xs = reinterpret_cast<const bf16*>(p.x) + (size_t)b * HWp * C;

    // per-channel affine of the GroupNorm (biased variance, eps inside gn_ab): v = x * scale + shift; zero beyond C
    for (int c = tid; c < 96; c += NT) {
        float sc = 0.f, sh = 0.f;
        if (c < C) {
            const int g = c / (C / p.G);
            const float a = p.gn_ab[((size_t)b * p.G + g) * 2], am = p.gn_ab[((size_t)b * p.G + g) * 2 + 1];
            const float gm = p.gamma[c];
            sc = a * gm;
            sh = p.beta[c] - am * gm;
        }
        tab[2 * c] = sc;
        tab[2 * c + 1] = sh;
    }
    // this thread always handles quarter dq = tid & 3 of a pixel: the shortcut weights of its 24 channels (8 per chunk) live in registers
    const int dq = tid & 3;
    float wn[DF_CH][3][8];
#if DS_BOUNDS
#pragma unroll
    for (int c = 0; c < DF_CH; ++c)
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = c * 32 + dq * 8 + j;
                wn[c][o][j] = ch < C ? p.wnin[o * C + ch] : 0.f;
            }
#else
    {
        // 18 range-checked 16-byte loads with arithmetic out-of-range offsets (C % 8 == 0: a group of 8 channels is inside or outside as a
        // whole) instead of 72 exec-masked single loads
        const rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wnin), (short)0, 3 * C * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < DF_CH; ++c)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int ch = c * 32 + dq * 8;
                const unsigned off = ((unsigned)(o * C + ch) * 4u) | ((unsigned)(ch >= C) << 31);
                const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs_n, (int)off, 0, 0), hi = __builtin_amdgcn_raw_buffer_load_b128(rs_n, (int)(off + 16u), 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wn[c][o][j] = __uint_as_float(lo[j]);
                    wn[c][o][4 + j] = __uint_as_float(hi[j]);
                }
            }
    }
#endif
    __syncthreads();                                                    // the table

    // ---- staging pass: raw values -> shortcut dot products; normalised + swish values -> the three chunk images
    const int lds_h = (tid >> 2) * PSTR + ((dq ^ (((tid >> 4) & 1) << 1)) << 4);
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
        const int slot = tid + it * NT, hp = slot >> 2;
        const int hr = hp / HCP, hc = hp - hr * HCP;
        const int hi = h0 + hr - 1, wi = w0 + hc - 1;
        const bool inside = hp < NPX && hc < TW + 2 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const bf16* const px = xs + (size_t)(inside ? hi * p.W + wi : 0) * C;
        u32x4 raw[DF_CH];
#pragma unroll
        for (int c = 0; c < DF_CH; ++c) {
            const bool ok = inside && c * 4 + dq < npc;
            const u32x4 v = *reinterpret_cast<const u32x4*>(px + (ok ? c * 32 + dq * 8 : 0));
            raw[c] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
        float dot[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DF_CH; ++c) {
            const bool ok = inside && c * 4 + dq < npc;
            float xv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[2 * e] = __uint_as_float(raw[c][e] << 16);
                xv[2 * e + 1] = __uint_as_float(raw[c][e] & 0xffff0000u);
            }
            const float* const tb = tab + 2 * (c * 32 + dq * 8);
            bf16x8 o8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = 0; o < 3; ++o) dot[o] = fmaf(xv[j], wn[c][o][j], dot[o]);
                const float v = fmaf(xv[j], tb[2 * j], tb[2 * j + 1]);
                o8[j] = (bf16)(ok ? swish_f(v) : 0.f);                  // zero padding applies to the convolution's INPUT (after the activation)
            }
            *reinterpret_cast<u32x4*>(smem + c * IMG + lds_h + it * 64 * PSTR) = __builtin_bit_cast(u32x4, o8);
        }
        // the four lanes of a pixel (dq = 0..3) are adjacent: two xor-shuffles complete the 1x1 shortcut of this pixel
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            dot[o] += __shfl_xor(dot[o], 1, 64);
            dot[o] += __shfl_xor(dot[o], 2, 64);
        }
        if (dq == 0 && inside && hr >= 1 && hr <= TH && hc >= 1 && hc <= TW)
            *reinterpret_cast<f32x4*>(nin + ((hr - 1) * TW + (hc - 1)) * 4) = f32x4{dot[0], dot[1], dot[2], 0.f};
    }
    // ---- this wave's 27 weight fragments (chunk-major tiles [cc*9 + tap][16][32]) straight into registers, then one barrier
    const char* const wbase = reinterpret_cast<const char*>(p.w3);
    bf16x8 wf[DF_CH][9];
#pragma unroll
    for (int c = 0; c < DF_CH; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wf[c][t] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(c * 9 + t) * 1024 + m * 64 + q * 16);
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
    __syncthreads();
    f32x4 acc[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < DF_CH; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t % 3;
#pragma unroll
            for (int i = 0; i < XT; ++i) {
                int a = xb[i];
                if (tx == 0 && ty == 1) a ^= 32;
                if (tx == 1) a ^= (ty == 1 ? (xm1 ^ 32) : xm1);
                if (tx == 2) a ^= (ty == 1 ? (xm2 ^ 32) : xm2);
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(smem + c * IMG + a + (ty * HCP + tx) * PSTR);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][t], xf, acc[i], 0, 0, 0);      // D^T = W . X^T
            }
        }
    // ---- epilogue: lane group q = 0 holds output channels 0..3 of pixel (tile i, lane m)
    if (q == 0) {
        float bs[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) bs[o] = p.b3[o] + p.bnin[o];
        float* const ob = p.out + (size_t)b * 3 * HWp;
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            int row_l, col_l;
            if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
            else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
            else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }
            const int ho = h0 + row_l, wo = w0 + col_l;
            if (ho < p.H && wo < p.W) {
                const f32x4 nv = *reinterpret_cast<const f32x4*>(nin + (row_l * TW + col_l) * 4);
                const float y0 = acc[i][0] + nv[0] + bs[0], y1 = acc[i][1] + nv[1] + bs[1], y2 = acc[i][2] + nv[2] + bs[2];
                const size_t pix = (size_t)ho * p.W + wo;
                ob[pix] = y0 > 20.f ? y0 : log1pf(expf(y0));            // F.softplus (beta 1, threshold 20)
                ob[(size_t)HWp + pix] = tanhf(y1);
                ob[2 * (size_t)HWp + pix] = tanhf(y2);
            }
        }
    }
}

}  // namespace

// x [B][H][W][C] bf16 (C % 8 == 0, C <= 96), gn_ab [B][G][2] = (rstd, rstd * mean) of GroupNorm(G, C) over x, gamma / beta [C],
// w3 = the 3x3 weight (3 outputs) packed chunk-major for 16 output rows and Cin padded to 96 (ds_pack_conv_weight: cin_pad 96, cout_pad 16,
// k_order 1), b3 [3], wnin [3][C] fp32 (nin_shortcut), bnin [3]; out [B][3][H][W] fp32 = softplus / tanh / tanh of the block's output.
extern "C" int ds_dec_final(const void* x, int B, int H, int W, int C, const float* gn_ab, int G, const float* gamma, const float* beta,
                            const void* w3, const float* b3, const float* wnin, const float* bnin, float* out, void* stream) {
    DS_REQUIRE(x && gn_ab && gamma && beta && w3 && b3 && wnin && bnin && out && B > 0 && H > 0 && W > 0, "dec_final: bad args");
    DS_REQUIRE(C > 0 && C % 8 == 0 && C <= 96 && G > 0 && C % G == 0, "dec_final: C=%d must be a multiple of 8, at most 96, and divisible by G=%d", C, G);
    if (!ds_aligned16(x) || !ds_aligned16(w3) || !ds_aligned16(wnin)) DS_FAIL(DS_EALIGN, "dec_final: x / w3 / wnin must be 16-byte aligned");
    ds_dec_final_params p{x, B, H, W, C, G, gn_ab, gamma, beta, w3, b3, wnin, bnin, out};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DS_REQUIRE(W > 8, "dec_final: images at most 8 wide are not supported (three 28 KB chunk images do not fit two blocks per CU)");
    int twl = 4;
    while ((1 << twl) < W && twl < 5) ++twl;
    const int TW = 1 << twl, TH = BM >> twl;
    dim3 grid(((H + TH - 1) / TH) * ((W + TW - 1) / TW), 1, B);
    if (twl == 5) {
        DS_SET_MAX_LDS(dec_final_kernel<5>, DF<5>::LDS, "dec_final<32>");
        hipLaunchKernelGGL(dec_final_kernel<5>, grid, dim3(NT), DF<5>::LDS, st, p);
    } else {
        DS_SET_MAX_LDS(dec_final_kernel<4>, DF<4>::LDS, "dec_final<16>");
        hipLaunchKernelGGL(dec_final_kernel<4>, grid, dim3(NT), DF<4>::LDS, st, p);
    }
    DS_CHECK_LAUNCH("dec_final");
    return DS_OK;
}
