// 3x3 stride-1 pad-1 convolution with 80 -> 80 channels (bf16 NHWC) of the VQGAN decoder's 256 x 128 stage, as a whole ResnetBlock body:
//     out = x + conv3x3(act(GroupNorm(G, 80)(x)))                (VQGAN.py:223-244 with temb = None and no nin_shortcut)
// reference: model/VAE VQGAN.py ResnetBlock / Decoder (tail row).  Sibling of convt4x4_c80.hip — same K decomposition:
// a K step of 32 = two (tap, 16-channel group) pairs, 9 taps x 5 groups = 45 pairs = 23 steps (the 46th pair has zero weights), the pixel
// fragment of lane (pixel m, k group kq) is 16 contiguous bytes of a staged halo pixel.  On the generic implicit GEMM this block was three
// launches: GroupNorm apply + swish (0.35 ms: read + write of the 335 MB tensor), the convolution with the residual (0.71 ms at 341 TF).
//   input   : GroupNorm affine + swish / ReLU applied to the loaded chunks on their way into LDS (optional), zero padding after it
//   block   : persistent, 4 waves, output tile 4 rows x 32 columns (wave = one row = 2 pixel tiles x 5 channel tiles); the 23 x 5 weight
//             fragments (115 KB) stay in LDS for the block's whole run of tiles, ONE halo image of 6 x 34 pixels x 176 bytes beside them;
//             the halo of tile t+1 and the residual pixels of tile t are requested before the MFMAs of tile t, a barrier releases the
//             image, the halo is written, the outputs leave, a second barrier publishes the image
//   MFMA    : mfma(W, X): a lane owns one pixel and 4 consecutive channels per channel tile; accumulators start from the bias; the residual
//             (the RAW x) comes back as 8-byte loads in the store pattern
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int C8_C = 80, C8_NG = 5, C8_NJ = 5, C8_NP = 9 * C8_NG, C8_KS = (C8_NP + 1) / 2;     // 45 pairs, 23 K steps
constexpr int C8_TW = 32, C8_TH = 4, C8_NT = 256;
constexpr int C8_HR = C8_TH + 2, C8_HC = C8_TW + 2, C8_PIXB = 176, C8_QPP = 10;
constexpr int C8_HBYTES = C8_HR * C8_HC * C8_PIXB;              // 35904
constexpr int C8_WBYTES = C8_KS * C8_NJ * 1024;                 // 117760
constexpr int C8_OFF_H = C8_WBYTES, C8_LDS = C8_OFF_H + C8_HBYTES;          // 153664: one block per CU
constexpr int C8_NPX = C8_HR * C8_HC;                           // 204 halo pixels
constexpr int C8_LT = 250, C8_PPI = C8_LT / C8_QPP;             // 250 loader threads: (pixel of 25, chunk tid % 10)
constexpr int C8_LIT = (C8_NPX + C8_PPI - 1) / C8_PPI;          // 9 load iterations
static_assert(C8_LDS <= 160 * 1024, "one block per CU");

typedef __amdgpu_buffer_rsrc_t c8_rsrc_t;
typedef __bf16 c8_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned c8_u32x2 __attribute__((ext_vector_type(2)));

struct C8Params {
    const void* x;      // [B][H][W][80] bf16
    const void* wpk;    // [23][5][64][8] bf16: ds_pack_conv3x3_c80
    const float* bias;  // [80] or null
    void* out;          // [B][H][W][80] bf16
    const float* gn_ab; const float* gamma; const float* beta; int G, act;   // optional act(GroupNorm(G, 80)(x)) on load; act: DS_ACT_*
    int add_x;          // 1: out += x (the block's residual)
    const void* res;    // residual source when it is not the convolution's input (ds_conv3x3_c80_res: x is the activated copy), else null
    float* stats_ws;    // optional [B][rps * 4][80][2]: per-channel (sum, sum of squares) of the output per (run, wave) — ds_gn_stats_finish
    int B, H, W, tiles_w, tiles_h, rps, per;       // runs per sample (a run = one block's tiles: never across samples), tiles per run
};

__device__ __forceinline__ float c8_act(float v, int act) {
    if (act == DS_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DS_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
    return v;
}

__global__ __launch_bounds__(C8_NT, 1) void conv3x3_c80_kernel(const C8Params p) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int tps = p.tiles_w * p.tiles_h, rb = blockIdx.x / p.rps, li = blockIdx.x - rb * p.rps;
    const int t0 = rb * tps + min(tps, li * p.per), t1 = rb * tps + min(tps, (li + 1) * p.per);
    const int nt = t1 - t0;
    float* const sws = p.stats_ws ? p.stats_ws + ((size_t)rb * (p.rps * 4) + li * 4 + wave) * C8_C * 2 : nullptr;
    if (nt <= 0) {                                             // (an empty run still owns its slots of the statistics)
        if (sws && m == 0)
#pragma unroll
            for (int j = 0; j < C8_NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2, DS_BX_STATS, 0.f);
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2 + 1, DS_BX_STATS, 0.f);
                }
        return;
    }
    // ---- weight fragments -> LDS (once; batches of loads)
    {
        constexpr int NV = C8_WBYTES / 16, WIT = (NV + C8_NT - 1) / C8_NT, NB = 4, PER = (WIT + NB - 1) / NB;
        const u32x4* src = reinterpret_cast<const u32x4*>(p.wpk);
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            u32x4 wst[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) wst[k] = DS_LD(u32x4, src + min(tid + (bt * PER + k) * C8_NT, NV - 1), DS_BX_W);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = tid + (bt * PER + k) * C8_NT;
                if (i < NV) *reinterpret_cast<u32x4*>(sm + i * 16) = wst[k];
            }
        }
    }
    f32x4 bv[C8_NJ];
#pragma unroll
    for (int j = 0; j < C8_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? DS_LD(float, p.bias + 16 * j + 4 * kq + r, DS_BX_BIAS) : 0.f;

    struct Tile { const char* base; c8_rsrc_t rs; const char* rbase; c8_rsrc_t rs_r; int b, i0, j0; };
    auto locate = [&](int t) {
        Tile r;
        const int per_b = p.tiles_w * p.tiles_h;
        r.b = t / per_b;
        const int q = t - r.b * per_b, th = q / p.tiles_w;
        r.i0 = th * C8_TH;
        r.j0 = (q - th * p.tiles_w) * C8_TW;
        r.base = reinterpret_cast<const char*>(p.x) + (size_t)r.b * p.H * p.W * C8_C * 2;
        r.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r.base), (short)0, p.H * p.W * C8_C * 2, 0x00020000);
        r.rbase = p.res ? reinterpret_cast<const char*>(p.res) + (size_t)r.b * p.H * p.W * C8_C * 2 : r.base;
        r.rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r.rbase), (short)0, p.H * p.W * C8_C * 2, 0x00020000);
        return r;
    };
    const int lq = tid % C8_QPP, lp = tid / C8_QPP;
    const bool loader = tid < C8_LT;
    int s_rc[C8_LIT], s_lds[C8_LIT];
#pragma unroll
    for (int it = 0; it < C8_LIT; ++it) {
        const int hp = min(it * C8_PPI + lp, C8_NPX - 1), hr = hp / C8_HC, hc = hp - hr * C8_HC;
        s_rc[it] = (hr << 8) | hc;
        s_lds[it] = hp * C8_PIXB + lq * 16;
    }
    u32x4 hv[C8_LIT];
    auto issue_halo = [&](const Tile& t) {
#pragma unroll
        for (int it = 0; it < C8_LIT; ++it) {
            const int hr = s_rc[it] >> 8, hc = s_rc[it] & 0xff;
            const int ih = t.i0 - 1 + hr, iw = t.j0 - 1 + hc;
            // (arithmetic out-of-range offsets: bit 31 = beyond the buffer; cut to 28 bits first — see conv7x7_c4.hip)
            const unsigned bad = (unsigned)(!loader) | (unsigned)(it * C8_PPI + lp >= C8_NPX) | (unsigned)((unsigned)ih >= (unsigned)p.H) | (unsigned)((unsigned)iw >= (unsigned)p.W);
            const unsigned off = (((unsigned)((ih * p.W + iw) * C8_C + lq * 8) * 2u) & 0x0fffffffu) | (bad << 31);
#if DS_BOUNDS
            if (bad || !ds_bx_ok(t.base + off, DS_BX_SRC0, 16)) { hv[it] = u32x4{0u, 0u, 0u, 0u}; continue; }
#endif
            hv[it] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, (int)off, 0, 0);
        }
    };
    // GroupNorm affine of this thread's 8 channels for sample b (see convt4x4_c80.hip): v = act(x * sc + sh), zero padding after it
    float gsc[8], gsh[8];
    const bool gn = p.gn_ab != nullptr;
    auto load_gn = [&](int b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = lq * 8 + e, g = c / (C8_C / p.G);
            const float a = DS_LD(float, p.gn_ab + ((size_t)b * p.G + g) * 2, DS_BX_GNAB), am = DS_LD(float, p.gn_ab + ((size_t)b * p.G + g) * 2 + 1, DS_BX_GNAB);
            const float gm = DS_LD(float, p.gamma + c, DS_BX_AUX0);
            gsc[e] = a * gm;
            gsh[e] = DS_LD(float, p.beta + c, DS_BX_AUX1) - am * gm;
        }
    };
    auto fill_halo = [&](char* h, const Tile& t) {
#pragma unroll
        for (int it = 0; it < C8_LIT; ++it) {
            u32x4 v = hv[it];
            if (gn) {                                          // (block-uniform)
                const int hr = s_rc[it] >> 8, hc = s_rc[it] & 0xff;
                const int ih = t.i0 - 1 + hr, iw = t.j0 - 1 + hc;
                const float keep = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? 1.f : 0.f;
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                    o8[2 * e] = (bf16)(keep * c8_act(fmaf(lo, gsc[2 * e], gsh[2 * e]), p.act));
                    o8[2 * e + 1] = (bf16)(keep * c8_act(fmaf(hi, gsc[2 * e + 1], gsh[2 * e + 1]), p.act));
                }
                v = __builtin_bit_cast(u32x4, o8);
            }
            if (loader && (it + 1 < C8_LIT || it * C8_PPI + lp < C8_NPX)) *reinterpret_cast<u32x4*>(h + s_lds[it]) = v;
        }
    };

    char* const himg = sm + C8_OFF_H;
    Tile cur = locate(t0);
    issue_halo(cur);
    if (gn) load_gn(cur.b);
    fill_halo(himg, cur);
    __syncthreads();

    // pixel fragment of K step ks: pair pidx = 2 ks + (kq >> 1) = (tap dy, dx; group g); the 46th pair (zero weights) re-reads pair 44
    int xoff[C8_KS];
#pragma unroll
    for (int ks = 0; ks < C8_KS; ++ks) {
        const int pidx = min(2 * ks + (kq >> 1), C8_NP - 1), tap = pidx / C8_NG, g = pidx - tap * C8_NG;
        xoff[ks] = ((tap / 3) * C8_HC + (tap % 3)) * C8_PIXB + (g * 16 + (kq & 1) * 8) * 2;
    }
    const int xb = (wave * C8_HC + m) * C8_PIXB;                // pixel tile i (= column half) of this wave's row: + 16 i * C8_PIXB
    float st1[C8_NJ][4], st2[C8_NJ][4];                        // statistics of this lane's 20 output channels (fp32 values before rounding)
#pragma unroll
    for (int j = 0; j < C8_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) st1[j][e] = st2[j][e] = 0.f;
    const char* const wl = sm + lane * 16;                     // weight fragment (ks, j): + (ks * C8_NJ + j) * 1024
    auto tile_body = [&](const int u, auto more_t) {
        constexpr bool more = decltype(more_t)::value;
        Tile nxt = cur;
        if constexpr (more) {
            nxt = locate(t0 + u + 1);
            issue_halo(nxt);
        }
        // residual pixels of THIS tile (raw x), in the store pattern: channels 16 j + 4 kq .. + 3 of pixel (row, 16 i + m)
        const int r = cur.i0 + wave;
        c8_u32x2 rx[2][C8_NJ];
        if (p.add_x) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = cur.j0 + 16 * i + m;
                const unsigned bad = (unsigned)(r >= p.H) | (unsigned)(c >= p.W);
                const unsigned o = ((unsigned)((r * p.W + c) * C8_C + 4 * kq) * 2u) & 0x0fffffffu;
#pragma unroll
                for (int j = 0; j < C8_NJ; ++j) {
#if DS_BOUNDS
                    if (bad || !ds_bx_ok(cur.rbase + o + 32u * j, p.res ? DS_BX_RES : DS_BX_SRC0, 8)) { rx[i][j] = c8_u32x2{0u, 0u}; continue; }
#endif
                    rx[i][j] = __builtin_bit_cast(c8_u32x2, __builtin_amdgcn_raw_buffer_load_b64(cur.rs_r, (int)((o + 32u * j) | (bad << 31)), 0, 0));
                }
            }
        }
        f32x4 acc[2][C8_NJ];
#pragma unroll
        for (int ks = 0; ks < C8_KS; ++ks) {
            bf16x8 xf[2], wf[C8_NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(himg + xb + 16 * i * C8_PIXB + xoff[ks]);
#pragma unroll
            for (int j = 0; j < C8_NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wl + (ks * C8_NJ + j) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < C8_NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], ks == 0 ? bv[j] : acc[i][j], 0, 0, 0);
        }
        if constexpr (more) {
            __syncthreads();                                   // every wave is done reading the halo image
            if (gn && nxt.b != cur.b) load_gn(nxt.b);          // (block-uniform, once per sample)
            fill_halo(himg, nxt);
        }
        // ---- (+ residual) bf16, five 8-byte stores per pixel
        bf16* const outb = reinterpret_cast<bf16*>(p.out) + (size_t)cur.b * p.H * p.W * C8_C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = cur.j0 + 16 * i + m;
            if (r < p.H && c < p.W) {
                bf16* o = outb + ((size_t)r * p.W + c) * C8_C + 4 * kq;
#pragma unroll
                for (int j = 0; j < C8_NJ; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                    if (p.add_x) {
                        v[0] += __uint_as_float(rx[i][j][0] << 16);
                        v[1] += __uint_as_float(rx[i][j][0] & 0xffff0000u);
                        v[2] += __uint_as_float(rx[i][j][1] << 16);
                        v[3] += __uint_as_float(rx[i][j][1] & 0xffff0000u);
                    }
                    c8_bf16x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o4[e] = (bf16)v[e];
                        st1[j][e] += v[e];
                        st2[j][e] = fmaf(v[e], v[e], st2[j][e]);
                    }
                    DS_ST(c8_bf16x4, reinterpret_cast<c8_bf16x4*>(o + 16 * j), DS_BX_OUT, o4);
                }
            }
        }
        if constexpr (more) __syncthreads();                   // the next halo image is complete
        cur = nxt;
    };
    for (int u = 0; u + 1 < nt; ++u) tile_body(u, std::true_type{});
    tile_body(nt - 1, std::false_type{});
    if (sws) {                                                 // sum over the 16 pixel lanes of a k group, lane m = 0 writes its 20 channels
#pragma unroll
        for (int j = 0; j < C8_NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = st1[j][e], b2 = st2[j][e];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    a += __shfl_xor(a, d, 64);
                    b2 += __shfl_xor(b2, d, 64);
                }
                if (m == 0) {
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2, DS_BX_STATS, a);
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2 + 1, DS_BX_STATS, b2);
                }
            }
    }
}

// w [Cout = 80][Cin = 80][3][3] fp32 (Conv2d layout) -> [ks][j][lane = kg * 16 + row][8] bf16: row of tile j = output channel 16 j + row;
// k slot kg * 8 + e = pair 2 ks + (kg >> 1) = (tap = 3 dy + dx, group g), input channel 16 g + 8 (kg & 1) + e; pair 45: zeros
__global__ void pack_conv3x3_c80_kernel(const float* w, bf16* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C8_KS * C8_NJ * 512) return;
    const int e = i & 7, lane = (i >> 3) & 63, j = (i >> 9) % C8_NJ, ks = i / (512 * C8_NJ);
    const int row = lane & 15, kg = lane >> 4;
    const int pidx = 2 * ks + (kg >> 1), tap = pidx / C8_NG, g = pidx - tap * C8_NG;
    const int ci = 16 * g + 8 * (kg & 1) + e, co = 16 * j + row;
    float v = 0.f;
    if (pidx < C8_NP) v = w[((size_t)co * C8_C + ci) * 9 + tap];
    dst[i] = (bf16)v;
}

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv3x3_c80(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" size_t ds_conv3x3_c80_weight_elems(void) { return (size_t)C8_KS * C8_NJ * 512; }

extern "C" int ds_pack_conv3x3_c80(const float* w, int Cout, int Cin, void* dst, void* stream) {
    DS_REQUIRE(w && dst, "pack_conv3x3_c80: null pointer");
    DS_REQUIRE(Cin == C8_C && Cout == C8_C, "pack_conv3x3_c80: %d -> %d unsupported (80 -> 80)", Cin, Cout);
    const int n = C8_KS * C8_NJ * 512;
    hipLaunchKernelGGL(pack_conv3x3_c80_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, reinterpret_cast<bf16*>(dst));
    DS_CHECK_LAUNCH("pack_conv3x3_c80");
    return DS_OK;
}

static void c8_partition(int B, int H, int W, int& tiles_w, int& tiles_h, int& rps, int& per) {
    tiles_w = (W + C8_TW - 1) / C8_TW;
    tiles_h = (H + C8_TH - 1) / C8_TH;
    const int tps = tiles_w * tiles_h;
    rps = 256 / B;                                             // persistent: about one block per CU, a block's tiles inside ONE sample
    if (rps < 1) rps = 1;
    if (rps > tps) rps = tps;
    per = (tps + rps - 1) / rps;
}

// slots of per-channel statistics partials per sample that ds_conv3x3_c80 writes (stats_ws = [B][slots][80][2] floats)
extern "C" int ds_conv3x3_c80_stats_slots(int B, int H, int W) {
    int tw, th, rps, per;
    c8_partition(B > 0 ? B : 1, H, W, tw, th, rps, per);
    return rps * 4;
}

static int c80_launch(const void* x, const void* res, int B, int H, int W, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                      const float* gamma, const float* beta, int act, int add_x, float* stats_ws, void* stream) {
    DS_REQUIRE(x && wpk && out, "conv3x3_c80: null pointer");
    DS_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_c80: bad sizes (B %d, %d x %d)", B, H, W);
    DS_REQUIRE(!gn_ab || (gamma && beta && G > 0 && C8_C % G == 0), "conv3x3_c80: the fused GroupNorm needs gamma, beta and a group count dividing 80 (G = %d)", G);
    DS_REQUIRE(act == DS_ACT_NONE || act == DS_ACT_RELU || act == DS_ACT_SILU, "conv3x3_c80: activation %d", act);
    DS_REQUIRE(x != out, "conv3x3_c80: in-place is not supported (neighbouring tiles read the input halo)");
    DS_REQUIRE((long long)H * W * C8_C * 2 < (1ll << 28), "conv3x3_c80: a sample must stay below 256 MB");
    if (!ds_aligned16(x) || !ds_aligned16(wpk) || !ds_aligned16(out)) DS_FAIL(DS_EALIGN, "conv3x3_c80: pointers must be 16-byte aligned");
    C8Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.out = out;
    p.gn_ab = gn_ab; p.gamma = gamma; p.beta = beta; p.G = gn_ab ? G : 1; p.act = gn_ab ? act : DS_ACT_NONE;
    p.add_x = (add_x || res) ? 1 : 0;
    p.res = res;
    p.stats_ws = stats_ws;
    p.B = B; p.H = H; p.W = W;
    c8_partition(B, H, W, p.tiles_w, p.tiles_h, p.rps, p.per);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV3X3_C80);
        h.set(DS_BX_SRC0, x, (long long)B * H * W * C8_C * 2);
        h.set(DS_BX_RES, res, (long long)B * H * W * C8_C * 2);
        h.set(DS_BX_W, wpk, (long long)C8_WBYTES);
        h.set(DS_BX_BIAS, bias, bias ? C8_C * 4 : 0);
        h.set(DS_BX_GNAB, gn_ab, gn_ab ? (long long)B * G * 2 * 4 : 0);
        h.set(DS_BX_AUX0, gamma, gn_ab ? C8_C * 4 : 0);
        h.set(DS_BX_AUX1, beta, gn_ab ? C8_C * 4 : 0);
        h.set(DS_BX_OUT, out, (long long)B * H * W * C8_C * 2);
        h.set(DS_BX_STATS, stats_ws, stats_ws ? (long long)B * p.rps * 4 * C8_C * 2 * 4 : 0);
        h.publish(st);
    }
#endif
    DS_SET_MAX_LDS(conv3x3_c80_kernel, C8_LDS, "conv3x3_c80");
    hipLaunchKernelGGL(conv3x3_c80_kernel, dim3(B * p.rps), dim3(C8_NT), C8_LDS, st, p);
    DS_CHECK_LAUNCH("conv3x3_c80");
    return DS_OK;
}

extern "C" int ds_conv3x3_c80(const void* x, int B, int H, int W, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                              const float* gamma, const float* beta, int act, int add_x, float* stats_ws, void* stream) {
    return c80_launch(x, nullptr, B, H, W, wpk, bias, out, gn_ab, G, gamma, beta, act, add_x, stats_ws, stream);
}

// out = res + conv3x3(h) + bias for an input h that already IS act(GroupNorm(res)) (written by ds_gn_apply): the block as two launches.  r04: with
// the norm + swish applied while the halo is staged the kernel is bound by that arithmetic on a 1.59 x redundant halo and one wave per SIMD
// (637 us at 64 x 256 x 128 x 80); the plain convolution takes 377 us and the apply pass ~110.
extern "C" int ds_conv3x3_c80_res(const void* h, const void* res, int B, int H, int W, const void* wpk, const float* bias, void* out, float* stats_ws,
                                  void* stream) {
    DS_REQUIRE(res && res != out && ds_aligned16(res), "conv3x3_c80_res: bad residual pointer");
    return c80_launch(h, res, B, H, W, wpk, bias, out, nullptr, 0, nullptr, nullptr, DS_ACT_NONE, 1, stats_ws, stream);
}
