// Last block of the VQGAN decoder in ONE kernel (bf16 activations; gfx950).
//
// Decoder.forward ends with ResnetBlock(80 -> 3) and the output activations (VQGAN.py:177-244, 390-398):
//     y = conv3x3(swish(GroupNorm16(x))) + nin_shortcut_1x1(x);   out = [softplus(y0), tanh(y1), tanh(y2)]
// on the largest tensor of the tail (512 x 256 x 80 per clip: 1.3 GB bf16 at batch 64).  As separate launches this stage read that tensor
// three times and wrote it once more (GroupNorm apply 0.42 ms, the two 80 -> 3 convolutions on the generic small-N tile 0.9 ms each, the
// activation kernel): 2.3 of the tail's 9.3 ms.  Here the tensor is read ONCE from HBM: a block stages the 34 x 10 halo of its 32 x 8 pixel tile
// for all 80 channels (three 32-channel chunk images in LDS, the swizzled 64-byte rows of conv3x3_halo3; channels 80..95 zero), applying the
// per-(sample, group) GroupNorm affine and the swish on the way; the 3x3 runs as in conv3x3_smalln (27 weight fragments in registers,
// 108 ds_read_b128 + 108 MFMAs per wave, one barrier) and the 1x1 shortcut on the RAW values rides the same accumulators (six more MFMAs per
// pixel tile, x fragments re-read from L2); the epilogue adds the biases, applies softplus / tanh and stores fp32 NCHW planes.
#include "common.hpp"
#include "conv_halo3_common.hpp"

namespace {

constexpr int DF_CH = 3;                                    // 32-channel chunk images (Cin <= 96)

template <int TWL>
struct DF {
    using G = HG<TWL>;
    static constexpr int IMG = G::H_IT * 64 * PSTR;         // one chunk image (rows of HCP pixels, as in conv3x3_halo3)
    static constexpr int OFF_TAB = DF_CH * IMG;             // [96] (scale, shift) fp32
    static constexpr int LDS = OFF_TAB + 96 * 8;
    static_assert(LDS <= 81920, "two blocks per CU");
};

struct ds_dec_final_params {
    const void* x; int B, H, W, C, G;
    const float* gn_ab; const float* gamma; const float* beta;
    const void* w3; const float* b3; const float* wnin; const float* bnin;
    float* out;
};

__device__ __forceinline__ float swish_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v)); }

// r04 (the first form: 1.2 - 1.35 ms at 64 x 512 x 256 x 80, VALU-bound in its staging pass):
//   * staging walks the REAL halo pixels and the REAL 16-byte channel pieces: thread = (pixel of the iteration, piece), NT / NPC pixels per
//     iteration, the piece fixed per thread (its 8 scales / shifts in registers) — 14 iterations of 8 values at C = 80 against 6 x 3 chunks
//     of 8 (the 36-pixel row pitch, the 384-slot store iterations and channels 80..95 were staged like data: 45 % of the lanes' work);
//   * the 1x1 shortcut left the staging loop (3 fma per value, 72 weight registers, two shuffles and an LDS round trip per pixel): it is six
//     more MFMAs per pixel tile in the 3x3 chain — W_nin as hi + lo bf16 rows of a 16-row A operand (fp32-exact weights), the raw x fragments
//     straight from global memory (L2: the block has just read them).
template <int TWL, int NPC>
__global__ __launch_bounds__(NT, DS_MINBLK) void dec_final_kernel(const ds_dec_final_params p) {
    using G = HG<TWL>;
    using D = DF<TWL>;
    constexpr int TW = G::TW, TH = G::TH, HCP = G::HCP, IMG = D::IMG;
    constexpr int HW2 = TW + 2, NHP = (TH + 2) * HW2;                    // real halo pixels
    constexpr int PPI = NT / NPC, SIT = (NHP + PPI - 1) / PPI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + D::OFF_TAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int tiles_w = (p.W + TW - 1) >> TWL;
    const int gx = gridDim.x, nwg = gx * gridDim.z;
    int wid = blockIdx.x + gx * blockIdx.z;
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);      // XCD-chunked order: neighbouring tiles on one L2
    const int bx = wid % gx, b = wid / gx;
    const int th = bx / tiles_w, tw = bx - th * tiles_w, h0 = th * TH, w0 = tw * TW;
    const int C = p.C, HWp = p.H * p.W, npc = C >> 3;                   // 16-byte pieces per pixel
    const bf16* const xs = reinterpret_cast<const bf16*>(p.x) + (size_t)b * HWp * C;
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(xs), (short)0, (int)((size_t)HWp * C * 2), 0x00020000);

    // ---- staging pass, part 1: the halo's raw pieces are requested before anything else (their HBM round trip covers the table set-up)
    const int sp = tid / NPC, oct = tid - sp * NPC;
    const bool s_act = sp < PPI && oct < npc;                            // (NPC = 12 serves every C <= 96: pieces beyond C are written as zeros)
    const int chunk = oct >> 2, dq = oct & 3;
    u32x4 raw[SIT];
    int hpl[SIT];                                                        // LDS pixel index (row pitch HCP) or -1
    unsigned msk[SIT];
    {
        int hr = sp / HW2, hc = sp - hr * HW2;                           // halo coordinates of this thread's pixel, advanced by PPI per iteration
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int hi = h0 + hr - 1, wi = w0 + hc - 1;
            const bool live = sp < PPI && hr < TH + 2;
            const bool inside = live && s_act && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned off = (unsigned)(((hi * p.W + wi) * C + oct * 8) * 2);
            raw[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((off & 0x7fffffffu) | ((unsigned)!inside << 31)), 0, 0);
            hpl[it] = live ? hr * HCP + hc : -1;
            msk[it] = inside ? 0xffffffffu : 0u;
            hc += PPI % HW2;
            hr += PPI / HW2;
            if (hc >= HW2) { hc -= HW2; ++hr; }
        }
    }
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
    // shortcut weights as A-operand rows (row m = output channel, rows >= 3 zero), hi and lo bf16 parts of the fp32 values
    bf16x8 wnh[DF_CH], wnl[DF_CH];
    {
        const rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wnin), (short)0, 3 * C * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < DF_CH; ++c) {
            const int ch = c * 32 + q * 8;
            const unsigned off = ((unsigned)(m * C + ch) * 4u) | ((unsigned)(m >= 3 || ch >= C) << 31);
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs_n, (int)off, 0, 0), bq = __builtin_amdgcn_raw_buffer_load_b128(rs_n, (int)(off + 16u), 0, 0);
            const float v[8] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]),
                                __uint_as_float(bq[0]), __uint_as_float(bq[1]), __uint_as_float(bq[2]), __uint_as_float(bq[3])};
            u32x4 hi, lo;
            ds_split8(v, hi, lo);
            wnh[c] = __builtin_bit_cast(bf16x8, hi);
            wnl[c] = __builtin_bit_cast(bf16x8, lo);
        }
    }
    __syncthreads();                                                    // the table

    // ---- staging pass, part 2: swish(GroupNorm(x)) -> the three chunk images
    {
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(tab + 16 * oct + 4 * j);
            sc[2 * j] = t4[0]; sh[2 * j] = t4[1]; sc[2 * j + 1] = t4[2]; sh[2 * j + 1] = t4[3];
        }
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            unsigned o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = __uint_as_float(raw[it][e] << 16), x1 = __uint_as_float(raw[it][e] & 0xffff0000u);
                const float v0 = fmaf(x0, sc[2 * e], sh[2 * e]), v1 = fmaf(x1, sc[2 * e + 1], sh[2 * e + 1]);
                const ds_f32x2 s2 = {swish_f(v0), swish_f(v1)};
                // zero padding applies to the convolution's INPUT (after the activation): pixels outside the image and pieces beyond C are zeros
                o4[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(s2, ds_bf16x2)) & msk[it];
            }
            if (hpl[it] >= 0) {
                const int hp = hpl[it];
                char* const row = smem + chunk * IMG + hp * PSTR;
                const int sw = ((hp >> 2) & 1) << 1;
                *reinterpret_cast<u32x4*>(row + ((dq ^ sw) << 4)) = u32x4{o4[0], o4[1], o4[2], o4[3]};
                if (NPC == 10 && oct >= 8) *reinterpret_cast<u32x4*>(row + (((dq + 2) ^ sw) << 4)) = u32x4{0u, 0u, 0u, 0u};   // channels 80..95
            }
        }
    }
    // ---- this wave's 27 weight fragments (chunk-major tiles [cc*9 + tap][16][32]) straight into registers, the raw x fragments of the
    //      shortcut (pixel (tile i, lane m), channels 8 q .. 8 q + 7 of each chunk), then one barrier
    const char* const wbase = reinterpret_cast<const char*>(p.w3);
    bf16x8 wf[DF_CH][9];
#pragma unroll
    for (int c = 0; c < DF_CH; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wf[c][t] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(c * 9 + t) * 1024 + m * 64 + q * 16);
    int xb[XT];
    u32x4 xn[XT][DF_CH];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
        else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }
        const int hp0 = row_l * HCP + col_l;
        xb[i] = hp0 * PSTR + ((q ^ (((hp0 >> 2) & 1) << 1)) << 4);
        const int ho = h0 + row_l, wo = w0 + col_l;
        const bool in_img = ho < p.H && wo < p.W;
#pragma unroll
        for (int c = 0; c < DF_CH; ++c) {
            const bool ok = in_img && c * 4 + q < npc;
            const unsigned off = (unsigned)(((ho * p.W + wo) * C + c * 32 + q * 8) * 2);
            xn[i][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((off & 0x7fffffffu) | ((unsigned)!ok << 31)), 0, 0);
        }
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
    // the shortcut on the raw values: (W_hi + W_lo) . x^T into the same accumulators
#pragma unroll
    for (int c = 0; c < DF_CH; ++c)
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            const bf16x8 xf = __builtin_bit_cast(bf16x8, xn[i][c]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wnh[c], xf, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wnl[c], xf, acc[i], 0, 0, 0);
        }
    // ---- epilogue: lane group 0 holds output channels 0..3 of pixel (tile i, lane m) for all four tiles; the activations (libm softplus /
    //      tanh: ~150 instructions) are spread over the lane groups — group q takes tile q — instead of running four times on a quarter wave
    {
        float y[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float t = __shfl(acc[i][o], m, 64);
                y[o] = q == i ? t : y[o];
            }
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (q >> 1); col_l = 16 * (q & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + q; col_l = m; }
        else { row_l = 8 * wave + q + 4 * (m >> 3); col_l = m & 7; }
        const int ho = h0 + row_l, wo = w0 + col_l;
        if (ho < p.H && wo < p.W) {
            const float y0 = y[0] + p.b3[0] + p.bnin[0], y1 = y[1] + p.b3[1] + p.bnin[1], y2 = y[2] + p.b3[2] + p.bnin[2];
            float* const ob = p.out + (size_t)b * 3 * HWp;
            const size_t pix = (size_t)ho * p.W + wo;
            ob[pix] = y0 > 20.f ? y0 : log1pf(expf(y0));                // F.softplus (beta 1, threshold 20)
            ob[(size_t)HWp + pix] = tanhf(y1);
            ob[2 * (size_t)HWp + pix] = tanhf(y2);
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
    // thread map of the staging pass: 10 pieces per pixel at C = 80 (the production decoder); every other C <= 96 on the 12-piece map
#define DS_DF_LAUNCH(TWL_, NPC_)                                                                          \
    do {                                                                                                  \
        DS_SET_MAX_LDS((dec_final_kernel<TWL_, NPC_>), DF<TWL_>::LDS, "dec_final");                       \
        hipLaunchKernelGGL((dec_final_kernel<TWL_, NPC_>), grid, dim3(NT), DF<TWL_>::LDS, st, p);         \
    } while (0)
    if (twl == 5) {
        if (C == 80) DS_DF_LAUNCH(5, 10);
        else DS_DF_LAUNCH(5, 12);
    } else {
        if (C == 80) DS_DF_LAUNCH(4, 10);
        else DS_DF_LAUNCH(4, 12);
    }
#undef DS_DF_LAUNCH
    DS_CHECK_LAUNCH("dec_final");
    return DS_OK;
}
