// ConvTranspose2d(C, C, 4, 2, 1) with C = 80 channels (bf16 NHWC): the VQGAN decoder's last Upsample, 256 x 128 -> 512 x 256 per clip.
// reference: model/VAE VQGAN.py Decoder: the `up` layers = nn.ConvTranspose2d(cur, nxt, 4, 2, 1) (tail row).
//
// 80 channels fit none of the 32-channel-chunk kernels (the four-tap halo kernel wants chunks in sixes, the generic implicit GEMM pads
// K = 4 x 80 to 12 steps and gathers every operand per tap through registers: 1.38 ms = 311 TF for 0.27 ms worth of output bytes).
// Here a K step of 32 is TWO (tap, 16-channel group) pairs: 4 taps x 5 groups = 20 pairs = 10 steps exactly, and the pixel fragment of
// lane (pixel m, k group kq) is 16 contiguous bytes of a staged halo pixel: pair 2 ks + (kq >> 1), channels 16 g + 8 (kq & 1) .. + 7.
//   phase   : output pixel (2i + py, 2j + px) = sum over a, b in {0, 1} of x[i + py - 1 + a][j + px - 1 + b] . w[.][.][3 - py - 2a][3 - px - 2b];
//             a block works on ONE phase (its 10 x 5 weight fragments, 50 KB, stay in LDS) and walks input tiles of 4 x 32 pixels; the four
//             phase blocks of a tile run are neighbours on one XCD (they read the same input through one L2)
//   input   : optionally relu(GroupNorm(G, 80)(x)) (the decoder's Normalize + ReLU in front of the layer): the per-(sample, group) affine is
//             applied to the loaded chunks on their way into LDS — the separate GroupNorm-apply pass (read + write of the input) disappears
//   LDS     : weights 50 KB (100 KB at 160 input channels) + ONE halo image of 5 x 33 pixels x 176 (336) bytes — the pad keeps the 16 lanes of a
//             ds_read_b128 on distinct banks; the halo of tile t+1 is requested before the MFMAs of tile t (range-checked buffer loads with
//             arithmetic out-of-range offsets), a barrier releases the image, the halo is written, the output stores leave, a second barrier
//             publishes it.  80 -> 80: two such blocks per CU (79 KB each)
//   MFMA    : mfma(W, X), rows = output channels, columns = pixels: a lane owns one pixel and 4 consecutive channels per channel tile
//             (5 x 8-byte stores; the four lanes of a pixel cover 32 contiguous bytes per store), accumulators start from the bias
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int U8_CO = 80, U8_NJ = U8_CO / 16;                   // output channels: 5 channel tiles of 16
constexpr int U8_TW = 32, U8_NT = 256;
// NGI = input channels / 16 (5: 80 -> 80, the last Upsample; 10: 160 -> 80, the first), TH = tile rows (a wave = TH / 4 rows), DB = two halo
// images (one barrier per tile) or one (two barriers: the 160-channel halo + 100 KB of weights leave room for one)
template <int NGI, int TH_, bool DB_, int BPC_ = 1>
struct U8 {
    static constexpr int BPC = BPC_;                                 // blocks per CU (BPC x LDS <= 160 KB, 256 registers per lane at 2)
    static constexpr int CI = 16 * NGI, KS = 4 * NGI / 2, TH = TH_, RPW = TH_ / 4, NPT = 2 * RPW;
    static constexpr bool DB = DB_;
    static constexpr int HR = TH + 1, HC = U8_TW + 1, PIXB = CI * 2 + 16, QPP = CI * 2 / 16;   // halo (TH + 1) x 33 pixels; the 16-byte pad keeps the
    static constexpr int HBYTES = HR * HC * PIXB;                                               // 16 lanes of a ds_read_b128 on distinct banks
    static constexpr int WBYTES = KS * U8_NJ * 1024;                 // per phase
    static constexpr int OFF_H = WBYTES, LDS = OFF_H + (DB ? 2 : 1) * HBYTES;
    static constexpr int NPX = HR * HC;
    static constexpr int PPI = U8_NT / QPP, LT = PPI * QPP;         // loader threads: thread = (pixel of PPI, chunk tid % QPP) — one chunk index per thread
    static constexpr int LIT = (NPX + PPI - 1) / PPI;
    static_assert(LDS * BPC <= 160 * 1024, "BPC blocks per CU");
    static_assert((PIXB / 4) % 4 == 0 && ((PIXB / 4) / 4) % 2 == 1, "pixel pitch: an odd number of 16-byte slots");
};
using U8A = U8<5, 4, false, 2>;   // 80 -> 80: 50 KB of weights + 29 KB, TWO blocks per CU — independent blocks overlap each other's MFMA, staging
                                  // and store phases (one 4-wave block per CU with 8-row tiles and two halo images: 0.78 ms against 0.55)
using U8B = U8<10, 4, false>;     // 160 -> 80: 100 KB of weights + 55 KB

typedef __amdgpu_buffer_rsrc_t u8_rsrc_t;
typedef unsigned u8_u32x2 __attribute__((ext_vector_type(2)));

struct U8Params {
    const void* x;      // [B][H][W][80] bf16
    const void* wpk;    // [4 phases][10][5][64][8] bf16: ds_pack_convt4x4_c80
    const float* bias;  // [80] or null
    void* out;          // [B][2H][2W][80] bf16
    const float* gn_ab; const float* gamma; const float* beta; int G;   // optional: relu(GroupNorm(G, 80)(x)) applied on the way into LDS
    float* stats_ws;    // optional [B][rps * 16][80][2]: per-channel (sum, sum of squares) of the output per (run, phase, wave) — ds_gn_stats_finish
    int B, H, W, tiles_w, tiles_h, rps, per, runs;     // runs per sample (a run's tiles stay inside ONE sample), tiles per run, runs = B * rps
};

template <typename M>
__global__ __launch_bounds__(U8_NT, M::BPC) void convt4x4_c80_kernel(const U8Params p) {
    constexpr int U8_C = M::CI, U8_NG = U8_C / 16, U8_KS = M::KS, U8_TH = M::TH, U8_HC = M::HC, U8_PIXB = M::PIXB, U8_QPP = M::QPP,
                  U8_HBYTES = M::HBYTES, U8_WBYTES = M::WBYTES, U8_OFF_H = M::OFF_H, U8_NPX = M::NPX, U8_LT = M::LT, U8_PPI = M::PPI, U8_LIT = M::LIT,
                  RPW = M::RPW, NPT = M::NPT;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    // block -> (phase, tile run): hardware block L runs on XCD L % 8; the four phases of a run sit on one XCD
    const int L = blockIdx.x;
    const int phase = (L >> 3) & 3, run = (L & 7) + 8 * (L >> 5);
    const int py = phase >> 1, px = phase & 1;
    if (run >= p.runs) return;
    const int tps = p.tiles_w * p.tiles_h, rb = run / p.rps, li = run - rb * p.rps;
    const int t0 = rb * tps + min(tps, li * p.per), t1 = rb * tps + min(tps, (li + 1) * p.per);
    const int nt = t1 - t0;
    float* const sws = p.stats_ws ? p.stats_ws + ((size_t)rb * (p.rps * 16) + (li * 4 + phase) * 4 + wave) * U8_CO * 2 : nullptr;
    if (nt <= 0) {                                             // (an empty run still owns its slots of the statistics)
        if (sws && m == 0)
#pragma unroll
            for (int j = 0; j < U8_NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2, DS_BX_STATS, 0.f);
                    DS_ST(float, sws + (16 * j + 4 * kq + e) * 2 + 1, DS_BX_STATS, 0.f);
                }
        return;
    }
    // ---- this phase's weight fragments -> LDS (once; two batches of loads)
    {
        constexpr int NV = U8_WBYTES / 16, WIT = (NV + U8_NT - 1) / U8_NT, HALF = (WIT + 1) / 2;
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.wpk) + (size_t)phase * U8_WBYTES);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 wst[HALF];
#pragma unroll
            for (int k = 0; k < HALF; ++k) wst[k] = DS_LD(u32x4, src + min(tid + (half * HALF + k) * U8_NT, NV - 1), DS_BX_W);
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
                const int i = tid + (half * HALF + k) * U8_NT;
                if (i < NV) *reinterpret_cast<u32x4*>(sm + i * 16) = wst[k];
            }
        }
    }
    f32x4 bv[U8_NJ];                                          // bias of this lane's rows of channel tile j: channels 16 j + 4 kq + r
#pragma unroll
    for (int j = 0; j < U8_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? DS_LD(float, p.bias + 16 * j + 4 * kq + r, DS_BX_BIAS) : 0.f;

    struct Tile { const char* base; u8_rsrc_t rs; int b, i0, j0; };
    auto locate = [&](int t) {
        Tile r;
        const int per_b = p.tiles_w * p.tiles_h;
        r.b = t / per_b;
        const int q = t - r.b * per_b, th = q / p.tiles_w;
        r.i0 = th * U8_TH;
        r.j0 = (q - th * p.tiles_w) * U8_TW;
        r.base = reinterpret_cast<const char*>(p.x) + (size_t)r.b * p.H * p.W * U8_C * 2;
        r.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r.base), (short)0, p.H * p.W * U8_C * 2, 0x00020000);
        return r;
    };
    // halo chunk slots of this thread: iteration it -> halo pixel it * 25 + tid / 10, always the 16-byte chunk q = tid % 10 of it (so that the
    // GroupNorm affine of its 8 channels lives in 16 registers); packed (row << 8 | col) and the LDS offset; threads 250 .. 255 carry none
    const int lq = tid % U8_QPP, lp = tid / U8_QPP;
    const bool loader = tid < U8_LT;
    int s_rc[U8_LIT], s_lds[U8_LIT];
#pragma unroll
    for (int it = 0; it < U8_LIT; ++it) {
        const int hp = min(it * U8_PPI + lp, U8_NPX - 1), hr = hp / U8_HC, hc = hp - hr * U8_HC;
        s_rc[it] = (hr << 8) | hc;
        s_lds[it] = hp * U8_PIXB + lq * 16;
    }
    u32x4 hv[U8_LIT];
    auto issue_halo = [&](const Tile& t) {
#pragma unroll
        for (int it = 0; it < U8_LIT; ++it) {
            const int hr = s_rc[it] >> 8, hc = s_rc[it] & 0xff, q = lq;
            const int ih = t.i0 + py - 1 + hr, iw = t.j0 + px - 1 + hc;
            // (arithmetic out-of-range offsets: bit 31 = beyond the buffer; cut to 28 bits first — see conv7x7_c4.hip)
            const unsigned bad = (unsigned)(!loader) | (unsigned)(it * U8_PPI + lp >= U8_NPX) | (unsigned)((unsigned)ih >= (unsigned)p.H) | (unsigned)((unsigned)iw >= (unsigned)p.W);
            const unsigned off = (((unsigned)((ih * p.W + iw) * U8_C + q * 8) * 2u) & 0x0fffffffu) | (bad << 31);
#if DS_BOUNDS
            if (bad || !ds_bx_ok(t.base + off, DS_BX_SRC0, 16)) { hv[it] = u32x4{0u, 0u, 0u, 0u}; continue; }
#endif
            hv[it] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, (int)off, 0, 0);
        }
    };
    // GroupNorm affine of this thread's 8 channels for sample b: v = relu(x * sc + sh); zero padding applies AFTER it (pixels outside the
    // image must stay 0: they are written as loaded — the range check returned zeros — and relu(sh) would not be 0)
    float gsc[8], gsh[8];
    const bool gn = p.gn_ab != nullptr;
    auto load_gn = [&](int b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = lq * 8 + e, g = c / (U8_C / p.G);
            const float a = DS_LD(float, p.gn_ab + ((size_t)b * p.G + g) * 2, DS_BX_GNAB), am = DS_LD(float, p.gn_ab + ((size_t)b * p.G + g) * 2 + 1, DS_BX_GNAB);
            const float gm = DS_LD(float, p.gamma + c, DS_BX_AUX0);
            gsc[e] = a * gm;
            gsh[e] = DS_LD(float, p.beta + c, DS_BX_AUX1) - am * gm;
        }
    };
    auto fill_halo = [&](char* h, const Tile& t) {
#pragma unroll
        for (int it = 0; it < U8_LIT; ++it) {
            u32x4 v = hv[it];
            if (gn) {                                          // (block-uniform)
                const int hr = s_rc[it] >> 8, hc = s_rc[it] & 0xff;
                const int ih = t.i0 + py - 1 + hr, iw = t.j0 + px - 1 + hc;
                const float keep = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? 1.f : 0.f;
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                    o8[2 * e] = (bf16)(keep * fmaxf(fmaf(lo, gsc[2 * e], gsh[2 * e]), 0.f));
                    o8[2 * e + 1] = (bf16)(keep * fmaxf(fmaf(hi, gsc[2 * e + 1], gsh[2 * e + 1]), 0.f));
                }
                v = __builtin_bit_cast(u32x4, o8);
            }
            if (loader && (it + 1 < U8_LIT || it * U8_PPI + lp < U8_NPX)) *reinterpret_cast<u32x4*>(h + s_lds[it]) = v;
        }
    };

    Tile cur = locate(t0);
    issue_halo(cur);
    if (gn) load_gn(cur.b);
    fill_halo(sm + U8_OFF_H, cur);
    __syncthreads();

    // pixel fragment of K step ks: pair pidx = 2 ks + (kq >> 1) = (tap a, b; group g), channels 16 g + 8 (kq & 1): byte offset inside the halo
    int xoff[U8_KS];
#pragma unroll
    for (int ks = 0; ks < U8_KS; ++ks) {
        const int pidx = 2 * ks + (kq >> 1), tap = pidx / U8_NG, g = pidx - tap * U8_NG;
        xoff[ks] = ((tap >> 1) * U8_HC + (tap & 1)) * U8_PIXB + (g * 16 + (kq & 1) * 8) * 2;
    }
    const int xb = ((RPW * wave) * U8_HC + m) * U8_PIXB;        // pixel tile i of this wave: + ((i >> 1) * U8_HC + 16 (i & 1)) * U8_PIXB
    float st1[U8_NJ][4], st2[U8_NJ][4];                        // statistics of this lane's 20 output channels (fp32 values before rounding)
#pragma unroll
    for (int j = 0; j < U8_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) st1[j][e] = st2[j][e] = 0.f;
    const char* const wl = sm + lane * 16;                     // weight fragment (ks, j): + (ks * U8_NJ + j) * 1024
    auto tile_body = [&](const int u, auto more_t) {
        constexpr bool more = decltype(more_t)::value;
        const char* const hcur = sm + U8_OFF_H + (M::DB ? (u & 1) * U8_HBYTES : 0);
        Tile nxt = cur;
        if constexpr (more) {
            nxt = locate(t0 + u + 1);
            issue_halo(nxt);
        }
        f32x4 acc[NPT][U8_NJ];
#pragma unroll
        for (int ks = 0; ks < U8_KS; ++ks) {
            bf16x8 xf[NPT], wf[U8_NJ];
#pragma unroll
            for (int i = 0; i < NPT; ++i)
                xf[i] = *reinterpret_cast<const bf16x8*>(hcur + xb + ((i >> 1) * U8_HC + 16 * (i & 1)) * U8_PIXB + xoff[ks]);
#pragma unroll
            for (int j = 0; j < U8_NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wl + (ks * U8_NJ + j) * 1024);
#pragma unroll
            for (int i = 0; i < NPT; ++i)
#pragma unroll
                for (int j = 0; j < U8_NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], ks == 0 ? bv[j] : acc[i][j], 0, 0, 0);
        }
        if constexpr (more) {
            if constexpr (!M::DB) __syncthreads();             // one halo image: every wave is done reading it
            if (gn && nxt.b != cur.b) load_gn(nxt.b);          // (block-uniform, once per sample)
            fill_halo(sm + U8_OFF_H + (M::DB ? ((u + 1) & 1) * U8_HBYTES : 0), nxt);      // (before the stores: see dwconv7_mfma2_kernel)
        }
        // ---- bf16, five 8-byte stores per pixel: lane = pixel m of each pixel tile, channels 16 j + 4 kq .. + 3
        bf16* const outb = reinterpret_cast<bf16*>(p.out) + (size_t)cur.b * (2 * p.H) * (2 * p.W) * U8_CO;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int r = cur.i0 + RPW * wave + (i >> 1), c = cur.j0 + 16 * (i & 1) + m;
            if (r < p.H && c < p.W) {
                bf16* o = outb + ((size_t)(2 * r + py) * (2 * p.W) + (2 * c + px)) * U8_CO + 4 * kq;
#pragma unroll
                for (int j = 0; j < U8_NJ; ++j) {
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    bf16x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (bf16)acc[i][j][e];
                        st1[j][e] += acc[i][j][e];
                        st2[j][e] = fmaf(acc[i][j][e], acc[i][j][e], st2[j][e]);
                    }
                    DS_ST(bf16x4_t, reinterpret_cast<bf16x4_t*>(o + 16 * j), DS_BX_OUT, v);
                }
            }
        }
        if constexpr (M::DB || more) __syncthreads();          // next halo image complete (two images: and every wave is done with this one)
        cur = nxt;
    };
    for (int u = 0; u + 1 < nt; ++u) tile_body(u, std::true_type{});
    tile_body(nt - 1, std::false_type{});
    if (sws) {                                                 // sum over the 16 pixel lanes of a k group, lane m = 0 writes its 20 channels
#pragma unroll
        for (int j = 0; j < U8_NJ; ++j)
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

// w [Cin][Cout = 80][4][4] fp32 (ConvTranspose2d layout) -> [phase][ks][j][lane = kg * 16 + row][8] bf16:
// row of tile j = output channel 16 j + row; k slot kg * 8 + e = pair 2 ks + (kg >> 1) = (tap, group g), input channel 16 g + 8 (kg & 1) + e,
// tap = 2 a + b -> kernel element (3 - py - 2 a, 3 - px - 2 b)
__global__ void pack_convt4x4_c80_kernel(const float* w, int Cin, bf16* dst) {
    const int NG = Cin / 16, KS = 4 * NG / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * KS * U8_NJ * 512) return;
    const int e = i & 7, lane = (i >> 3) & 63, j = (i >> 9) % U8_NJ, ks = (i / (512 * U8_NJ)) % KS, phase = i / (512 * U8_NJ * KS);
    const int row = lane & 15, kg = lane >> 4, py = phase >> 1, px = phase & 1;
    const int pidx = 2 * ks + (kg >> 1), tap = pidx / NG, g = pidx - tap * NG, a = tap >> 1, b = tap & 1;
    const int ci = 16 * g + 8 * (kg & 1) + e, co = 16 * j + row, kh = 3 - py - 2 * a, kw = 3 - px - 2 * b;
    dst[i] = (bf16)w[(((size_t)ci * U8_CO + co) * 4 + kh) * 4 + kw];
}

template <typename M>
static void u8_partition(int B, int H, int W, int& tiles_w, int& tiles_h, int& rps, int& per) {
    tiles_w = (W + U8_TW - 1) / U8_TW;
    tiles_h = (H + M::TH - 1) / M::TH;
    const int tps = tiles_w * tiles_h;
    rps = 64 * M::BPC / B;                                     // 64 runs x 4 phases = one block per CU (x BPC); a run's tiles inside ONE sample
    if (rps < 1) rps = 1;
    if (rps > tps) rps = tps;
    per = (tps + rps - 1) / rps;
}

template <typename M>
static int u8_launch(const void* x, int B, int H, int W, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                     const float* gamma, const float* beta, float* stats_ws, hipStream_t st) {
    U8Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.out = out;
    p.gn_ab = gn_ab; p.gamma = gamma; p.beta = beta; p.G = gn_ab ? G : 1;
    p.stats_ws = stats_ws;
    p.B = B; p.H = H; p.W = W;
    u8_partition<M>(B, H, W, p.tiles_w, p.tiles_h, p.rps, p.per);
    p.runs = B * p.rps;
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONVT4X4_C80);
        h.set(DS_BX_SRC0, x, (long long)B * H * W * M::CI * 2);
        h.set(DS_BX_W, wpk, (long long)4 * M::WBYTES);
        h.set(DS_BX_BIAS, bias, bias ? U8_CO * 4 : 0);
        h.set(DS_BX_GNAB, gn_ab, gn_ab ? (long long)B * G * 2 * 4 : 0);
        h.set(DS_BX_AUX0, gamma, gn_ab ? M::CI * 4 : 0);
        h.set(DS_BX_AUX1, beta, gn_ab ? M::CI * 4 : 0);
        h.set(DS_BX_OUT, out, (long long)B * 4 * H * W * U8_CO * 2);
        h.set(DS_BX_STATS, stats_ws, stats_ws ? (long long)B * p.rps * 16 * U8_CO * 2 * 4 : 0);
        h.publish(st);
    }
#endif
    // grid: block L = (run = (L & 7) + 8 (L >> 5), phase = (L >> 3) & 3); blocks beyond the last run return at once
    const int nb = 32 * ((p.runs + 7) / 8);
    DS_SET_MAX_LDS(convt4x4_c80_kernel<M>, M::LDS, "convt4x4_c80");
    hipLaunchKernelGGL(convt4x4_c80_kernel<M>, dim3(nb), dim3(U8_NT), M::LDS, st, p);
    DS_CHECK_LAUNCH("convt4x4_c80");
    return DS_OK;
}

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_convt4x4_c80(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" size_t ds_convt4x4_c80_weight_elems(int Cin) { return (size_t)4 * (4 * (Cin / 16) / 2) * U8_NJ * 512; }

extern "C" int ds_pack_convt4x4_c80(const float* w, int Cin, int Cout, void* dst, void* stream) {
    DS_REQUIRE(w && dst, "pack_convt4x4_c80: null pointer");
    DS_REQUIRE((Cin == 80 || Cin == 160) && Cout == U8_CO, "pack_convt4x4_c80: %d -> %d unsupported (80 or 160 -> 80)", Cin, Cout);
    const int n = (int)ds_convt4x4_c80_weight_elems(Cin);
    hipLaunchKernelGGL(pack_convt4x4_c80_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, Cin, reinterpret_cast<bf16*>(dst));
    DS_CHECK_LAUNCH("pack_convt4x4_c80");
    return DS_OK;
}

// slots of per-channel statistics partials per sample that ds_convt4x4_c80 writes (stats_ws = [B][slots][80][2] floats)
extern "C" int ds_convt4x4_c80_stats_slots(int B, int H, int W, int Cin) {
    int tw, th, rps, per;
    if (Cin == 160) u8_partition<U8B>(B > 0 ? B : 1, H, W, tw, th, rps, per);
    else u8_partition<U8A>(B > 0 ? B : 1, H, W, tw, th, rps, per);
    return rps * 16;
}

extern "C" int ds_convt4x4_c80(const void* x, int B, int H, int W, int Cin, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                               const float* gamma, const float* beta, float* stats_ws, void* stream) {
    DS_REQUIRE(x && wpk && out, "convt4x4_c80: null pointer");
    DS_REQUIRE(Cin == 80 || Cin == 160, "convt4x4_c80: %d input channels unsupported (80, 160)", Cin);
    DS_REQUIRE(!gn_ab || (gamma && beta && G > 0 && Cin % G == 0), "convt4x4_c80: the fused GroupNorm needs gamma, beta and a group count dividing %d (G = %d)", Cin, G);
    DS_REQUIRE(B > 0 && H > 0 && W > 0, "convt4x4_c80: bad sizes (B %d, %d x %d)", B, H, W);
    DS_REQUIRE((long long)H * W * Cin * 2 < (1ll << 28), "convt4x4_c80: a sample must stay below 256 MB");
    if (!ds_aligned16(x) || !ds_aligned16(wpk) || !ds_aligned16(out)) DS_FAIL(DS_EALIGN, "convt4x4_c80: pointers must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (Cin == 160) return u8_launch<U8B>(x, B, H, W, wpk, bias, out, gn_ab, G, gamma, beta, stats_ws, st);
    return u8_launch<U8A>(x, B, H, W, wpk, bias, out, gn_ab, G, gamma, beta, stats_ws, st);
}
