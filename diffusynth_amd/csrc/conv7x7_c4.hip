// 7x7 stride-1 pad-3 convolution of an image with at most FOUR real input channels (the U-Net's init_conv: 4 -> 96 at 256x64), bf16.
// reference: model/DiffSynth.py ConditionedUnet.init_conv = nn.Conv2d(channels, init_dim, 7, padding=3).
//
// The generic implicit-GEMM kernel gathers 49 taps x 8 padded channels per pixel (K = 392, a 16-byte global load per tap and pixel) and
// reaches 182 TF = 0.43 ms on the headline workload — 6x the 64 us it takes to write the 403 MB result.  With four channels a pixel is
// 8 bytes, so two horizontally adjacent taps of one pixel are 16 CONTIGUOUS bytes of a staged halo row: one K step of 32 = one kernel row
// dy = 8 horizontal taps (the 8th with zero weights) x 4 channels, and the B fragment of lane (pixel m, k group kq) is the 16 bytes at
// halo[row + dy][col + m + 2 kq] — one ds_read2_b64, no gather.  7 K steps instead of 13.
//   block   : 4 waves, output tile 8 rows x 32 columns; wave = 2 rows = 4 pixel tiles of 16 x all 96 channels (24 accumulator tiles);
//             persistent (two blocks per CU), a block walks a contiguous run of tiles
//   LDS     : the 7 x 6 weight fragments (42 KB, loaded once per block) + two halo images of 14 rows x 40 pixels x 8 bytes; the halo of tile
//             t+2 is requested before the MFMAs of tile t (range-checked buffer loads: pixels outside the image read as zeros), the halo of
//             tile t+1 is written after them, then the 4 x 3 output stores per lane of tile t leave — one barrier per tile
//   MFMA    : mfma(W, X): rows = output channels, columns = pixels, so a lane owns ONE pixel and — with the weight rows permuted at pack
//             time (row 4g + r of tile j = channel 32 (j >> 1) + 8 g + 4 (j & 1) + r) — three groups of 8 consecutive channels of it:
//             store k of the four lanes g of a pixel covers the 64 contiguous bytes of channels 32 k .. 32 k + 31, no LDS transpose
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int I7_TW = 32, I7_TH = 8, I7_NT = 256;
constexpr int I7_HW = 40, I7_HR = I7_TH + 6;                 // halo image: 14 rows x 40 pixels (38 used) x 8 bytes
constexpr int I7_HBYTES = I7_HR * I7_HW * 8;                 // 4480
constexpr int I7_WBYTES = 7 * 6 * 1024;                      // 43008: [dy][channel tile][lane] x 16 bytes
constexpr int I7_OFF_H = I7_WBYTES, I7_LDS = I7_OFF_H + 2 * I7_HBYTES;      // 51968: two (three) blocks per CU
constexpr int I7_NPX = I7_HR * 38;                           // 532 halo pixels
constexpr int I7_LIT = (I7_NPX + I7_NT - 1) / I7_NT;         // 3 load iterations

typedef __amdgpu_buffer_rsrc_t i7_rsrc_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct I7Params {
    const void* x;      // [B][H][W][Cx] bf16, Cx = 8 (channels 4..7 ignored) or 4
    const void* wpk;    // packed weights: ds_pack_conv7x7_c4
    const float* bias;  // [96] or null
    void* out;          // [B][H][W][96] bf16
    int B, H, W, Cx, tiles_w, tiles_h, ntiles;
};

// X3 = the split-precision form of the bf16x3 tier: x [B][H][W][4] FP32 is split into hi = bf16(x) / lo = bf16(x - hi) on its way to LDS (two halo
// planes per image), the weights arrive as hi / lo fragment sets, every (row, tile) product is three MFMAs (w_lo x_hi + w_hi x_lo + w_hi x_hi,
// small terms first) and the result leaves as fp32; 104 KB of LDS: one block per CU.
template <bool X3>
__global__ __launch_bounds__(I7_NT, X3 ? 1 : 2) void conv7x7_c4_kernel(const I7Params p) {
    constexpr int WB = X3 ? 2 * I7_WBYTES : I7_WBYTES;         // weight bytes in LDS (X3: hi set, then lo set)
    constexpr int HB = X3 ? 2 * I7_HBYTES : I7_HBYTES;         // one halo image (X3: hi plane, then lo plane)
    constexpr int OFF_H = WB;
    constexpr int PXB = X3 ? 16 : 0;                            // bytes per stored input pixel (bf16: Cx * 2 at run time)
    typedef typename std::conditional<X3, u32x4, u32x2_t>::type hv_t;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int per = (p.ntiles + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(p.ntiles, t0 + per);
    if (t0 >= t1) return;
    // ---- weights -> LDS (once; requested in two batches of loads, not one load -> wait -> write per loop iteration)
    {
        constexpr int NV = WB / 16, WIT = (NV + I7_NT - 1) / I7_NT, HALF = (WIT + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 wst[HALF];
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
                const int i = tid + (half * HALF + k) * I7_NT;
                wst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(p.wpk) + min(i, NV - 1), DS_BX_W);
            }
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
                const int i = tid + (half * HALF + k) * I7_NT;
                if (i < NV) *reinterpret_cast<u32x4*>(sm + i * 16) = wst[k];
            }
        }
    }
    f32x4 bv[6];                                               // bias of this lane's rows of channel tile j: the accumulators start from it
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? DS_LD(float, p.bias + 32 * (j >> 1) + 8 * kq + 4 * (j & 1) + r, DS_BX_BIAS) : 0.f;

    struct Tile { const char* base; i7_rsrc_t rs; int b, h0, w0; };
    auto locate = [&](int t) {
        Tile r;
        const int per_b = p.tiles_w * p.tiles_h;
        r.b = t / per_b;
        const int q = t - r.b * per_b, th = q / p.tiles_w;
        r.h0 = th * I7_TH;
        r.w0 = (q - th * p.tiles_w) * I7_TW;
        const int pxb = X3 ? PXB : p.Cx * 2;
        r.base = reinterpret_cast<const char*>(p.x) + (size_t)r.b * p.H * p.W * pxb;
        r.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r.base), (short)0, p.H * p.W * pxb, 0x00020000);
        return r;
    };
    // halo pixel slots of this thread: slot -> (row, column) of the 14 x 38 halo, LDS offset of its 8 bytes
    int s_r[I7_LIT], s_c[I7_LIT];
#pragma unroll
    for (int it = 0; it < I7_LIT; ++it) {
        const int s = tid + it * I7_NT;
        s_r[it] = s / 38;
        s_c[it] = s - s_r[it] * 38;
    }
    // Two halos are in flight: tile u+2's is requested at the start of tile u and written to LDS at the end of tile u+1 — one tile of
    // MFMAs (1.2 us) does not cover a memory round trip, two tiles and a barrier do.  The two register sets swap roles every tile, so the loop
    // is written out for two tiles (hvA / hvB are compile-time names: no indexed registers, no merged wait-count state between the copies).
    hv_t hvA[I7_LIT], hvB[I7_LIT];
    auto issue_halo = [&](hv_t (&hv)[I7_LIT], const Tile& t) {
#pragma unroll
        for (int it = 0; it < I7_LIT; ++it) {
            const int hi = t.h0 + s_r[it] - 3, wi = t.w0 + s_c[it] - 3;
            // (arithmetic, not a select on the data: an offset with bit 31 set is beyond the buffer; cut to 28 bits first so that bit 31
            // plus the offset plus 8 bytes cannot wrap around 2^32 into the buffer — a sample is far below 256 MB)
            const unsigned bad = (unsigned)(tid + it * I7_NT >= I7_NPX) | (unsigned)((unsigned)hi >= (unsigned)p.H) | (unsigned)((unsigned)wi >= (unsigned)p.W);
            const unsigned off = (((unsigned)(hi * p.W + wi) * (X3 ? (unsigned)PXB : (unsigned)p.Cx * 2u)) & 0x0fffffffu) | (bad << 31);
#if DS_BOUNDS
            if (bad || !ds_bx_ok(t.base + off, DS_BX_SRC0, X3 ? 16 : 8)) { hv[it] = hv_t{}; continue; }
#endif
            if constexpr (X3) hv[it] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, (int)off, 0, 0);
            else hv[it] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(t.rs, (int)off, 0, 0));
        }
    };
    auto fill_halo = [&](char* h, const hv_t (&hv)[I7_LIT]) {
#pragma unroll
        for (int it = 0; it < I7_LIT; ++it)
            if (it + 1 < I7_LIT || tid + it * I7_NT < I7_NPX) {
                char* const d = h + (s_r[it] * I7_HW + s_c[it]) * 8;
                if constexpr (X3) {
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    bf16x4_t vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = __uint_as_float(hv[it][e]);
                        vh[e] = (bf16)f;
                        vl[e] = (bf16)(f - (float)vh[e]);
                    }
                    *reinterpret_cast<u32x2_t*>(d) = __builtin_bit_cast(u32x2_t, vh);
                    *reinterpret_cast<u32x2_t*>(d + I7_HBYTES) = __builtin_bit_cast(u32x2_t, vl);
                } else {
                    *reinterpret_cast<u32x2_t*>(d) = hv[it];
                }
            }
    };

    const int nt = t1 - t0;
    Tile cur = locate(t0), nx1 = cur;
    issue_halo(hvA, cur);
    // columns 38, 39 of the halo rows are read (tap 7 of the last pixels, zero weights): keep them finite
    for (int i = tid; i < (X3 ? 4 : 2) * I7_HR * 2; i += I7_NT) {       // (every plane of both images: planes are I7_HBYTES apart)
        const int img = i / (I7_HR * 2), r = (i / 2) % I7_HR, c = 38 + (i & 1);
        *reinterpret_cast<u32x2_t*>(sm + OFF_H + img * I7_HBYTES + (r * I7_HW + c) * 8) = u32x2_t{0u, 0u};
    }
    fill_halo(sm + OFF_H, hvA);
    if (nt > 1) {
        nx1 = locate(t0 + 1);
        issue_halo(hvA, nx1);
    }
    __syncthreads();

    // B fragment (pixels) of pixel tile i, kernel row dy: 16 bytes at halo[(2 wave + (i >> 1)) + dy][16 (i & 1) + m + 2 kq]
    const int xoff = ((2 * wave) * I7_HW + m + 2 * kq) * 8;
    const char* const wl = sm + lane * 16;                     // A fragment (dy, j): + (dy * 6 + j) * 1024
    // tile u: hf holds tile u+1's halo (requested during tile u-1), hi is free and receives tile u+2's
    auto tile_body = [&](const int u, hv_t (&hf)[I7_LIT], hv_t (&hi)[I7_LIT], auto m1_t, auto m2_t) {
        constexpr bool m1 = decltype(m1_t)::value, m2 = decltype(m2_t)::value;
        const char* const hcur = sm + OFF_H + (u & 1) * HB;
        Tile nx2 = nx1;
        if constexpr (m2) {
            nx2 = locate(t0 + u + 2);
            issue_halo(hi, nx2);
        }
        f32x4 acc[4][6];
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) {
            bf16x8 xf[4], wf[6], xl[X3 ? 4 : 1], wlo[X3 ? 6 : 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* a = hcur + xoff + (((i >> 1) + dy) * I7_HW + 16 * (i & 1)) * 8;
                const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(a), hi2 = *reinterpret_cast<const u32x2_t*>(a + 8);
                xf[i] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi2[0], hi2[1]});
                if constexpr (X3) {
                    const u32x2_t l0 = *reinterpret_cast<const u32x2_t*>(a + I7_HBYTES), l1 = *reinterpret_cast<const u32x2_t*>(a + I7_HBYTES + 8);
                    xl[i] = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                wf[j] = *reinterpret_cast<const bf16x8*>(wl + (dy * 6 + j) * 1024);
                if constexpr (X3) wlo[j] = *reinterpret_cast<const bf16x8*>(wl + I7_WBYTES + (dy * 6 + j) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if constexpr (X3) {                         // small terms first
                        if (dy == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[j], xf[i], bv[j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[j], xf[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xl[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                    } else {
                        if (dy == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], bv[j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if constexpr (m1) fill_halo(sm + OFF_H + ((u + 1) & 1) * HB, hf);                 // (before the stores: see dwconv7_mfma2_kernel)
        // ---- bf16, three 16-byte stores per pixel: lane = pixel m of each pixel tile, channels 32 k + 8 kq .. + 7
        bf16* const outb = reinterpret_cast<bf16*>(p.out) + (size_t)cur.b * p.H * p.W * 96;
        float* const outf = reinterpret_cast<float*>(p.out) + (size_t)cur.b * p.H * p.W * 96;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = cur.h0 + 2 * wave + (i >> 1), w = cur.w0 + 16 * (i & 1) + m;
            if (h < p.H && w < p.W) {
                bf16* o = outb + ((size_t)h * p.W + w) * 96 + 8 * kq;
                if constexpr (X3) {                             // fp32: the four lanes of a pixel cover 128 contiguous bytes per k
                    float* of = outf + ((size_t)h * p.W + w) * 96 + 8 * kq;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        DS_ST(f32x4, reinterpret_cast<f32x4*>(of + 32 * k), DS_BX_OUT, acc[i][2 * k]);
                        DS_ST(f32x4, reinterpret_cast<f32x4*>(of + 32 * k + 4), DS_BX_OUT, acc[i][2 * k + 1]);
                    }
                } else
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = (bf16)acc[i][2 * k + (e >> 2)][e & 3];     // channel 32 k + 8 kq + e = tile 2 k + (e >> 2), register e & 3
                    }
                    DS_ST(bf16x8, reinterpret_cast<bf16x8*>(o + 32 * k), DS_BX_OUT, v);
                }
            }
        }
        __syncthreads();                                       // next halo image complete; every wave is done with this one
        cur = nx1;
        nx1 = nx2;
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int u = 0;
    for (; u + 3 < nt; u += 2) {
        tile_body(u, hvA, hvB, T, T);
        tile_body(u + 1, hvB, hvA, T, T);
    }
    const int r = nt - u;                                      // 1 .. 3 tiles left, tile u's successor in hvA
    if (r == 3) {
        tile_body(u, hvA, hvB, T, T);
        tile_body(u + 1, hvB, hvA, T, F);
        tile_body(u + 2, hvA, hvB, F, F);
    } else if (r == 2) {
        tile_body(u, hvA, hvB, T, F);
        tile_body(u + 1, hvB, hvA, F, F);
    } else {
        tile_body(u, hvA, hvB, F, F);
    }
}

// w [96][Cin <= 4][7][7] fp32 -> A fragments [dy][channel tile j][lane = kg * 16 + row][8]: row 4 g + r of tile j = channel 32 (j >> 1) + 8 g + 4 (j & 1) + r,
// k slot kg * 8 + e = horizontal tap 2 kg + (e >> 2), input channel e & 3 (tap 7 and channels >= Cin: zero)
__global__ void pack_conv7x7_c4_kernel(const float* w, int Cin, bf16* dst, int x3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 7 * 6 * 64 * 8) return;
    const int e = i & 7, lane = (i >> 3) & 63, j = (i >> 9) % 6, dy = i / (6 * 512);
    const int row = lane & 15, kg = lane >> 4;
    const int ch = 32 * (j >> 1) + 8 * (row >> 2) + 4 * (j & 1) + (row & 3), dx = 2 * kg + (e >> 2), ci = e & 3;
    float v = 0.f;
    if (dx < 7 && ci < Cin) v = w[((size_t)ch * Cin + ci) * 49 + dy * 7 + dx];
    const bf16 hi = (bf16)v;
    dst[i] = hi;
    if (x3) dst[7 * 6 * 64 * 8 + i] = (bf16)(v - (float)hi);     // the lo fragment set follows the hi set
}

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv7x7_c4(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" size_t ds_conv7x7_c4_weight_elems(void) { return (size_t)7 * 6 * 64 * 8; }

extern "C" int ds_pack_conv7x7_c4(const float* w, int Cout, int Cin, void* dst, void* stream) {
    DS_REQUIRE(w && dst, "pack_conv7x7_c4: null pointer");
    DS_REQUIRE(Cout == 96 && Cin >= 1 && Cin <= 4, "pack_conv7x7_c4: %d -> %d unsupported (<= 4 -> 96)", Cin, Cout);
    hipLaunchKernelGGL(pack_conv7x7_c4_kernel, dim3((7 * 6 * 64 * 8 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, Cin,
                       reinterpret_cast<bf16*>(dst), 0);
    DS_CHECK_LAUNCH("pack_conv7x7_c4");
    return DS_OK;
}

// split-precision form (bf16x3 tier): dst holds 2 x ds_conv7x7_c4_weight_elems() bf16 — the hi fragments, then the lo fragments
extern "C" int ds_pack_conv7x7_c4_x3(const float* w, int Cout, int Cin, void* dst, void* stream) {
    DS_REQUIRE(w && dst, "pack_conv7x7_c4_x3: null pointer");
    DS_REQUIRE(Cout == 96 && Cin >= 1 && Cin <= 4, "pack_conv7x7_c4_x3: %d -> %d unsupported (<= 4 -> 96)", Cin, Cout);
    hipLaunchKernelGGL(pack_conv7x7_c4_kernel, dim3((7 * 6 * 64 * 8 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, Cin,
                       reinterpret_cast<bf16*>(dst), 1);
    DS_CHECK_LAUNCH("pack_conv7x7_c4_x3");
    return DS_OK;
}

extern "C" int ds_conv7x7_c4(const void* x, int B, int H, int W, int Cx, const void* wpk, const float* bias, void* out, void* stream) {
    DS_REQUIRE(x && wpk && out, "conv7x7_c4: null pointer");
    DS_REQUIRE(B > 0 && H > 0 && W > 0 && (Cx == 4 || Cx == 8), "conv7x7_c4: bad sizes (B %d, %d x %d, %d stored channels)", B, H, W, Cx);
    DS_REQUIRE((long long)H * W * Cx * 2 < (1ll << 28), "conv7x7_c4: a sample must stay below 256 MB");
    if (!ds_aligned16(x) || !ds_aligned16(wpk) || !ds_aligned16(out)) DS_FAIL(DS_EALIGN, "conv7x7_c4: pointers must be 16-byte aligned");
    I7Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cx = Cx;
    p.tiles_w = (W + I7_TW - 1) / I7_TW;
    p.tiles_h = (H + I7_TH - 1) / I7_TH;
    p.ntiles = B * p.tiles_w * p.tiles_h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV7X7_C4);
        h.set(DS_BX_SRC0, x, (long long)B * H * W * Cx * 2);
        h.set(DS_BX_W, wpk, (long long)I7_WBYTES);
        h.set(DS_BX_BIAS, bias, bias ? 96 * 4 : 0);
        h.set(DS_BX_OUT, out, (long long)B * H * W * 96 * 2);
        h.publish(st);
    }
#endif
    const int nb = p.ntiles < 512 ? p.ntiles : 512;            // persistent: two blocks per CU
    DS_SET_MAX_LDS(conv7x7_c4_kernel<false>, I7_LDS, "conv7x7_c4");
    hipLaunchKernelGGL(conv7x7_c4_kernel<false>, dim3(nb), dim3(I7_NT), I7_LDS, st, p);
    DS_CHECK_LAUNCH("conv7x7_c4");
    return DS_OK;
}

// x [B][H][W][4] FP32, wpk from ds_pack_conv7x7_c4_x3, out [B][H][W][96] FP32 = conv7x7(x, padding 3) + bias in split precision
// (x_hi w_hi + x_lo w_hi + x_hi w_lo on bf16 MFMAs, fp32 accumulation: the init convolution of the bf16x3 tier)
extern "C" int ds_conv7x7_c4_x3(const float* x, int B, int H, int W, const void* wpk, const float* bias, float* out, void* stream) {
    DS_REQUIRE(x && wpk && out, "conv7x7_c4_x3: null pointer");
    DS_REQUIRE(B > 0 && H > 0 && W > 0, "conv7x7_c4_x3: bad sizes (B %d, %d x %d)", B, H, W);
    DS_REQUIRE((long long)H * W * 16 < (1ll << 28), "conv7x7_c4_x3: a sample must stay below 256 MB");
    if (!ds_aligned16(x) || !ds_aligned16(wpk) || !ds_aligned16(out)) DS_FAIL(DS_EALIGN, "conv7x7_c4_x3: pointers must be 16-byte aligned");
    I7Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cx = 4;
    p.tiles_w = (W + I7_TW - 1) / I7_TW;
    p.tiles_h = (H + I7_TH - 1) / I7_TH;
    p.ntiles = B * p.tiles_w * p.tiles_h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV7X7_C4);
        h.set(DS_BX_SRC0, x, (long long)B * H * W * 16);
        h.set(DS_BX_W, wpk, (long long)2 * I7_WBYTES);
        h.set(DS_BX_BIAS, bias, bias ? 96 * 4 : 0);
        h.set(DS_BX_OUT, out, (long long)B * H * W * 96 * 4);
        h.publish(st);
    }
#endif
    constexpr int LDS3 = 2 * I7_WBYTES + 4 * I7_HBYTES;         // 103936: one block per CU
    const int nb = p.ntiles < 256 ? p.ntiles : 256;
    DS_SET_MAX_LDS(conv7x7_c4_kernel<true>, LDS3, "conv7x7_c4_x3");
    hipLaunchKernelGGL(conv7x7_c4_kernel<true>, dim3(nb), dim3(I7_NT), LDS3, st, p);
    DS_CHECK_LAUNCH("conv7x7_c4_x3");
    return DS_OK;
}
