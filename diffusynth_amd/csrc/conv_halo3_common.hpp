// Shared pieces of the 16x16x32-MFMA halo convolution kernels (conv3x3_halo3.hip, conv_quad_halo3.hip): LDS layout, buffer
// load / store helpers, the packed-fp32 polynomial GELU and the branch-free register epilogue.  See conv3x3_halo3.hip for the
// design notes.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "conv_epilogue.hpp"

namespace {

constexpr int PSTR = 64;                       // LDS row = 32 bf16 channels of one pixel / one output channel
constexpr int BM = 256, BN = 96, NT = 256;
constexpr int XT = 4, WT = 6;                  // wave tile 64 px x 96 ch = 4 x 6 accumulators of 16 x 16
constexpr int HALO_BYTES = 448 * PSTR;         // 28672: 7 store iterations of 64 pixels (TW = 8: 34 x 12 = 408 halo pixels)
constexpr int B_BYTES = BN * PSTR;             // 6144
constexpr int B_STRIDE = B_BYTES + 64;         // + a 64-byte pad: target of the idle lanes of the 1.5-round tile store
constexpr int SHL_BYTES = 10 * BN * 4;         // shift table [9 border classes][BN] + one zero row (lanes without an output pixel)
constexpr int OFF_B = 0, OFF_SHL = 3 * B_STRIDE, OFF_H = OFF_SHL + SHL_BYTES;
constexpr int LDS_BYTES = OFF_H + 2 * HALO_BYTES;   // 79808 <= 81920: two blocks per CU
constexpr unsigned VOFF_NONE = 0x80000000u;
// Lane <-> channel map of a 96-channel N-block (r03, last hours).  Accumulator tile j of lane group g = lane >> 4 holds channels
//     EPI_CH(j) + 8 g + r   (r = accumulator register 0..3),   EPI_CH(j) = 32 (j >> 1) + 4 (j & 1)
// so the tile pair (2k, 2k + 1) of a lane is 8 CONSECUTIVE channels and the four lane groups of a pixel are 32 consecutive ones: a 16-byte bf16
// store / residual load instruction covers 64 contiguous bytes per pixel (fp32: two instructions, 128).  Until then a lane owned channels
// 24 g .. 24 g + 23 and every instruction was four separate 16-byte pieces per pixel, 48 bytes apart — four times the requests to L2, for the
// same instruction count.  The map costs nothing: weights sit in LDS in channel order, it is only the ROW a lane reads for its A fragment
// (W_ROW below) and the offsets of the epilogue.
__host__ __device__ constexpr int EPI_CH(int j) { return 32 * (j >> 1) + 4 * (j & 1); }
// LDS row (= output channel inside the N-block) of the A fragment of tile j for lane m = lane & 15: EPI_CH(j) + W_ROW0(m)
__host__ __device__ constexpr int W_ROW0(int m) { return 8 * (m >> 2) + (m & 3); }
// 16-byte-slot swizzle of a weight row: the four row groups a fragment read touches are 8 rows = 512 bytes apart (the same banks): XOR by the group
__host__ __device__ constexpr int W_SWZ(int row) { return (-((row >> 3) & 3)) & 3; }         // beyond any num_records: the buffer range check returns zeros

// GELU table of the bf16 epilogue: T(a) = a Phi(-a) at the midpoint of every bf16 bucket of |x| (sign dropped, 8 exponent + 7 mantissa
// bits = the upper half of the fp32 pattern) between 2^-12 and 8: 15 binades x 128 = 1920 floats
constexpr unsigned GELU_TAB_LO = 0x39800000u, GELU_TAB_HI = 0x40FF0000u;       // 2^-12 and 7.96875 (the last bucket below 8)
constexpr int GELU_TAB_BASE = GELU_TAB_LO >> 16, GELU_TAB_N = (GELU_TAB_HI >> 16) - GELU_TAB_BASE + 1;   // 1920
template <int TWL> struct HG {
    static constexpr int TW = 1 << TWL, TH = BM >> TWL;
    static constexpr int HCP = TW + 4;                      // halo row pitch in pixels (TW + 2 used): a multiple of 4
    static constexpr int NPX = (TH + 2) * HCP;              // 360 / 360 / 408
    static constexpr int H_IT = (NPX * 4 + NT - 1) / NT;    // 6 / 6 / 7 load-store iterations of 256 x 16 B
    static constexpr int HH0 = (H_IT + 1) / 2, HH1 = H_IT - HH0;   // halo refill in two halves
    static_assert(H_IT * 64 * PSTR <= HALO_BYTES, "halo store iterations must stay inside the buffer");
    // conv3x3_halo3 sizes its two halo buffers by the tile width (24 KB for the 32- and 16-wide tiles, 28 KB for the 8-wide one): the 9 KB
    // that frees hold the GELU table of the epilogue (below, 7680 bytes) with two blocks per CU still fitting
    static constexpr int HB = H_IT * 64 * PSTR;
    static constexpr bool LUT = H_IT <= 6;
    static constexpr int OFF_LUT = OFF_H + 2 * HB;
    static constexpr int LDS = OFF_LUT + (LUT ? GELU_TAB_N * 4 : 0);
    static_assert(LDS <= 81920, "two blocks per CU");
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ u32x4 buf_ld16(rsrc_t rs, const char* base, unsigned voff, unsigned soff, int bounds_buf) {
#if DS_BOUNDS
#ifndef DS_BX_SKIP_AUX
#define DS_BX_SKIP_AUX 0
#endif
    if (!(DS_BX_SKIP_AUX && (bounds_buf == DS_BX_AUX0 || bounds_buf == DS_BX_AUX1)))
    if (voff < VOFF_NONE && !ds_bx_ok(base + soff + voff, bounds_buf, 16)) return u32x4{0u, 0u, 0u, 0u};
#endif
    (void)base; (void)bounds_buf;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
}

constexpr int SG_MFMA = 0x8, SG_VMEM = 0x10, SG_DSR = 0x100, SG_DSW = 0x200;
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf-GELU for the bf16 epilogue without transcendentals: gelu(x) = relu(x) - T(|x|), T(a) = a * Phi(-a) (the Gaussian tail, zero to
// 1.5e-5 beyond a = 4.5).  T is a degree-11 polynomial in t = 2a/4.5 - 1 on [0, 4.5] (interpolation at the Chebyshev-Lobatto points,
// so T(0) = T(4.5) = 0 and clamping t at 1 continues it by zero); Horner on eight independent values.  Max |error| against
// 0.5 x (1 + erf(x / sqrt 2)) evaluated in fp32: 2.0e-5 (bf16 rounds a value of 0.01 by 4e-5).  16 v_fma-class instructions per value
// and no v_rcp / v_exp (27 cycles each beside a busy matrix pipe).
// Plain v_fma_f32, NOT v_pk_fma_f32 (the r02 form).  Measured on MI355X (tools/ubench/coissue.hip): beside a wave that keeps the
// SIMD's matrix pipe busy, a v_pk_fma_f32 / v_pk_add_f32 of the partner wave issues every 39 cycles, a v_fma_f32 every 8.4 (alone:
// 6.3 / 5.3) — packed fp32 is 2.3x MORE expensive per value exactly when the other block of the CU is in its K loop (same-box A/B of
// the whole step: +1.3 %); the library is built with -fno-slp-vectorize so that hipcc does not re-pack these chains.
__device__ __forceinline__ void gelu_poly8(float (&w)[8]) {
    constexpr float K = 2.0f / 4.5f;
    constexpr float c[12] = {2.748536319e-02f, -1.331737041e-01f, 2.467794865e-01f, -1.447154731e-01f, -2.043376267e-01f, 4.207932651e-01f,
                             -2.013681531e-01f, -1.442166418e-01f, 1.726166159e-01f, -1.050815172e-02f, -4.117569700e-02f, 1.182068978e-02f};
    float t[8], u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        t[e] = fmaf(fminf(fabsf(w[e]), 4.5f), K, -1.0f);
        u[e] = fmaf(c[11], t[e], c[10]);
    }
#pragma unroll
    for (int i = 9; i >= 0; --i)
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = fmaf(u[e], t[e], c[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = fmaxf(w[e], 0.0f) - u[e];
}

// The same function from a table in LDS (conv3x3_halo3, 32- and 16-wide tiles), indexed DIRECTLY by the upper 16 bits of |x| (clamped
// to [2^-12, 8)): T at the midpoint of that bf16 bucket, no interpolation.  5 VALU + one ds_read_b32 per value (clamp, shift, address,
// relu, subtract) instead of the polynomial's 16 or the 10 of an interpolated table — these kernels are bound by the epilogue's vector
// instruction count beside the other block's MFMA loop (8.4 cycles per instruction there, tools/ubench/coissue.hip).  Error against the
// exact function: <= 6.6e-4 absolute (at x = -2), 1.7e-4 rms over N(0, 2.5) inputs = 5 % of the bf16 rounding of the result that
// follows (3.2e-3 rms), mean 3e-7 (midpoints: unbiased); below 2^-12 the result is x/2 to within 1.2e-4.  oracle check: tests/test_hip_kernels.py.
__device__ __forceinline__ void gelu_tab8(float (&w)[8], const char* lut) {
    const char* const tb = lut - GELU_TAB_BASE * 4;             // (table entry of pattern t at lut + (t - BASE) * 4)
    float h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float a = __builtin_amdgcn_fmed3f(fabsf(w[k]), __uint_as_float(GELU_TAB_LO), __uint_as_float(GELU_TAB_HI));
        h[k] = *reinterpret_cast<const float*>(tb + ((__float_as_uint(a) >> 16) << 2));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = fmaxf(w[k], 0.0f) - h[k];
}

__device__ __forceinline__ void buf_st16(rsrc_t rs, const char* base, unsigned voff, u32x4 v, int bounds_buf) {
#if DS_BOUNDS
    if (voff >= VOFF_NONE || !ds_bx_ok(base + voff, bounds_buf, 16)) return;
#endif
    (void)base; (void)bounds_buf;
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)voff, 0, 0);
}

// ---- register epilogue: lane = pixel (lane & 15) of each of the wave's 4 pixel tiles, channels EPI_CH(j) + 8 * (lane >> 4) + r of accumulator tile j (three runs of 8 consecutive channels per lane).
// Branch-free: a lane without an output pixel (ragged tile) computes on a zero factor, the zero row of the shift table and a zero
// residual, and its stores / residual loads carry an out-of-range buffer offset (dropped / zeros by the range check) — the
// exec-masked version spent more time in s_and_saveexec / s_cbranch than in arithmetic (12 masked regions per wave tile).
// TAB: the GELU comes from the LDS table at `lut` (gelu_tab8) — a template parameter, not a runtime test of the pointer (the table's LDS
// address then folds into the gathers' immediate offset).  Issuing the gathers of group g+1 under the arithmetic of group g (hand
// software-pipelined, sched_barrier-fenced) measured the same: the partner wave already covers that latency.
template <int ACT, bool NCLS9, bool HAS_RES, bool TAB = false, typename CoordFn>
__device__ __forceinline__ void halo3_epilogue(const ds_conv_params& p, f32x4 (&acc)[XT][WT], int b, int n0, int outHW, const float* shl,
                                               CoordFn coord, float& s1, float& s2, float ga, int lane, const char* lut = nullptr) {
    const int g = lane >> 4, n_loc = 8 * g;                 // channels 32 k + 8 g .. + 7 of store k (EPI_CH)
    const unsigned sample_bytes = (unsigned)outHW * p.out_C * 2u;
    char* const obase = reinterpret_cast<char*>(p.out) + (size_t)b * sample_bytes;
    const char* const rbase = reinterpret_cast<const char*>(p.res) + (size_t)b * sample_bytes;
    const rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(obase, (short)0, (int)sample_bytes, 0x00020000);
    const rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(HAS_RES ? rbase : obase), (short)0, HAS_RES ? (int)sample_bytes : 0, 0x00020000);
    const int cout_v = (p.Cout + 7) / 8 * 8;
    const bool nine = (p.gn_ab != nullptr || p.gn_part != nullptr) && p.ncls == 9;
    // per pixel tile i: border-class row of the shift table, GroupNorm factor (0 for lanes without a pixel) and the three 16-byte store /
    // residual offsets — set up tile by tile, the residual vectors of tile i + 1 requested while tile i is computed (24 registers of
    // residuals in flight instead of 48 for the whole wave tile)
    unsigned voff[2][3];
    float gai[2];
    const float* shrow[2];
    u32x4 rres[HAS_RES ? 2 : 1][3];
    auto setup = [&](int i) {
        const ConvCoord c = coord(i);
        int cls = 0;
        if constexpr (NCLS9) cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
        if (!nine) cls = 0;                                    // bias-only table: one row
        shrow[i & 1] = shl + (c.ok ? cls : 9) * BN + n_loc;    // row 9 of the table is zero
        gai[i & 1] = c.ok ? ga : 0.f;
        // (arithmetic, not a select: the compiler turns the select into exec-masked branches, 12 masked regions per wave tile; an offset with
        // bit 31 set is beyond every buffer = VOFF_NONE)
        const unsigned o = ((unsigned)(c.pix * p.out_C + p.out_c0 + n0 + n_loc) * 2u) & 0x7fffffffu, nok = (unsigned)!c.ok;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            voff[i & 1][k] = (o + 64u * k) | ((nok | (unsigned)(n0 + n_loc + 32 * k >= cout_v)) << 31);
            if constexpr (HAS_RES) rres[i & 1][k] = (DS_EPI_ABL & 2) ? u32x4{0u, 0u, 0u, 0u} : buf_ld16(rs_r, rbase, voff[i & 1][k], 0u, DS_BX_RES);
        }
    };
    setup(0);
    // plain fp32 VALU (never packed: see gelu_poly8), four independent statistics chains
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        if (i + 1 < XT) setup(i + 1);
        const float gi = gai[i & 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                      // 8 channels = accumulator tiles 2k, 2k+1
            const f32x4 sa = *reinterpret_cast<const f32x4*>(shrow[i & 1] + 32 * k), sb = *reinterpret_cast<const f32x4*>(shrow[i & 1] + 32 * k + 4);
            const f32x4 a0 = acc[i][2 * k], a1 = acc[i][2 * k + 1];
            float w[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w[r] = fmaf(gi, a0[r], sa[r]);
                w[4 + r] = fmaf(gi, a1[r], sb[r]);
            }
            if constexpr (ACT == DS_ACT_GELU && !(DS_EPI_ABL & 4)) {
                if constexpr (TAB) gelu_tab8(w, lut);       // (the table exists for the 32- / 16-wide tiles)
                else gelu_poly8(w);
            }
            if constexpr (HAS_RES) {                       // bf16 -> fp32: the low / high half of each dword
                const u32x4 rr = rres[i & 1][k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w[2 * e] += __uint_as_float(rr[e] << 16);
                    w[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u);
                }
            }
            bf16x8 o8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1a += w[2 * e];
                s1b += w[2 * e + 1];
                s2a = fmaf(w[2 * e], w[2 * e], s2a);
                s2b = fmaf(w[2 * e + 1], w[2 * e + 1], s2b);
                o8[2 * e] = (bf16)w[2 * e];
                o8[2 * e + 1] = (bf16)w[2 * e + 1];
            }
            if constexpr (!(DS_EPI_ABL & 1)) buf_st16(rs_o, obase, voff[i & 1][k], __builtin_bit_cast(u32x4, o8), DS_BX_OUT);
        }
    }
    s1 += s1a + s1b;
    s2 += s2a + s2b;
}

// ---- the same epilogue with LINE-SIZED stores (r03).  Above, a 16-byte store instruction is 64 separate pieces (a lane owns one pixel and 24
// consecutive channels: its three pieces are 16 bytes apart, the four lane groups of a pixel 48 bytes apart, pixels a row apart), and so is a
// residual load: 64 L2 requests per kilobyte.  Here the 16 pixels x 96 channels of one accumulator tile row go through a wave-private LDS tile
// (the halo buffers are dead after the K loop; every wave is past the loop's last barrier) and leave as 192 consecutive 16-byte pieces, 12 per
// pixel: whole 128-byte lines per instruction where the block's 96 channels are the tensor's row (out_C = 96), 192-byte runs otherwise.
// Without a residual the tile is staged as bf16 (pitch 208 B, statistics taken in the accumulator layout as before); with one it is staged as
// fp32 (pitch 400 B) and the bf16 residual is loaded, added, rounded and counted on the contiguous side (the same fp32 sum, rounded once).
template <int ACT, bool NCLS9, bool HAS_RES, bool TAB = false, typename CoordFn, typename Coord2Fn>
__device__ __forceinline__ void halo3_epilogue_rows(const ds_conv_params& p, f32x4 (&acc)[XT][WT], int b, int n0, int outHW, const float* shl,
                                                    CoordFn coord, Coord2Fn coord2, char* stage, float& s1, float& s2, float ga, int lane,
                                                    const char* lut = nullptr) {
    constexpr int PITCH = HAS_RES ? 400 : 208;
    const int m = lane & 15, g = lane >> 4, n_loc = 8 * g;
    const unsigned sample_bytes = (unsigned)outHW * p.out_C * 2u;
    char* const obase = reinterpret_cast<char*>(p.out) + (size_t)b * sample_bytes;
    const char* const rbase = reinterpret_cast<const char*>(p.res) + (size_t)b * sample_bytes;
    const rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(obase, (short)0, (int)sample_bytes, 0x00020000);
    const rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(HAS_RES ? rbase : obase), (short)0, HAS_RES ? (int)sample_bytes : 0, 0x00020000);
    const int cout_v = (p.Cout + 7) / 8 * 8;
    const bool nine = (p.gn_ab != nullptr || p.gn_part != nullptr) && p.ncls == 9;
    int pxl[3], pq[3];                                       // contiguous side: piece lane + 64 t of the tile row = pixel pxl, channels 8 pq .. 8 pq + 7
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int pc = lane + 64 * t;
        pxl[t] = pc / 12;
        pq[t] = pc - 12 * pxl[t];
    }
    char* const wr = stage + m * PITCH + n_loc * (HAS_RES ? 4 : 2);          // + 32 channels per k
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        unsigned off[3];
        u32x4 rres[HAS_RES ? 3 : 1];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const ConvCoord c2 = coord2(i, pxl[t]);
            const unsigned bad = (unsigned)!c2.ok | (unsigned)(n0 + 8 * pq[t] >= cout_v);
            off[t] = (((unsigned)(c2.pix * p.out_C + p.out_c0 + n0 + 8 * pq[t]) * 2u) & 0x7fffffffu) | (bad << 31);
            if constexpr (HAS_RES) rres[t] = buf_ld16(rs_r, rbase, off[t], 0u, DS_BX_RES);
        }
        const ConvCoord c = coord(i);
        int cls = 0;
        if constexpr (NCLS9) cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
        if (!nine) cls = 0;
        const float* const shrow = shl + (c.ok ? cls : 9) * BN + n_loc;
        const float gi = c.ok ? ga : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x4 sa = *reinterpret_cast<const f32x4*>(shrow + 32 * k), sb = *reinterpret_cast<const f32x4*>(shrow + 32 * k + 4);
            const f32x4 a0 = acc[i][2 * k], a1 = acc[i][2 * k + 1];
            float w[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w[r] = fmaf(gi, a0[r], sa[r]);
                w[4 + r] = fmaf(gi, a1[r], sb[r]);
            }
            if constexpr (ACT == DS_ACT_GELU) {
                if constexpr (TAB) gelu_tab8(w, lut);
                else gelu_poly8(w);
            }
            if constexpr (HAS_RES) {
                *reinterpret_cast<f32x4*>(wr + 128 * k) = f32x4{w[0], w[1], w[2], w[3]};
                *reinterpret_cast<f32x4*>(wr + 128 * k + 16) = f32x4{w[4], w[5], w[6], w[7]};
            } else {
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1a += w[2 * e];
                    s1b += w[2 * e + 1];
                    s2a = fmaf(w[2 * e], w[2 * e], s2a);
                    s2b = fmaf(w[2 * e + 1], w[2 * e + 1], s2b);
                    o8[2 * e] = (bf16)w[2 * e];
                    o8[2 * e + 1] = (bf16)w[2 * e + 1];
                }
                *reinterpret_cast<u32x4*>(wr + 64 * k) = __builtin_bit_cast(u32x4, o8);
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if constexpr (HAS_RES) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + pxl[t] * PITCH + pq[t] * 32), hi = *reinterpret_cast<const f32x4*>(stage + pxl[t] * PITCH + pq[t] * 32 + 16);
                float w[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const u32x4 rr = rres[t];
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w[2 * e] += __uint_as_float(rr[e] << 16);
                    w[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u);
                    s1a += w[2 * e];
                    s1b += w[2 * e + 1];
                    s2a = fmaf(w[2 * e], w[2 * e], s2a);
                    s2b = fmaf(w[2 * e + 1], w[2 * e + 1], s2b);
                    o8[2 * e] = (bf16)w[2 * e];
                    o8[2 * e + 1] = (bf16)w[2 * e + 1];
                }
                buf_st16(rs_o, obase, off[t], __builtin_bit_cast(u32x4, o8), DS_BX_OUT);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(stage + pxl[t] * PITCH + pq[t] * 16);
                buf_st16(rs_o, obase, off[t], v, DS_BX_OUT);
            }
        }
    }
    s1 += s1a + s1b;
    s2 += s2a + s2b;
}

// ---- epilogue of the split-precision tier (ds_conv_params.flags, DS_CONV_F_OUT_*): the same lane layout, fp32 results stored either
// as TWO bf16 planes (hi = bf16(v) at channel n, lo = bf16(v - hi) at channel Cout + n of an image with 2 * Cout bf16 channels: the
// input format of the next split convolution) or as plain fp32 (with an optional fp32 residual: what the fp32 kernels around the
// 3x3 convolutions read).  Exact-erf GELU (gelu_fast: 1.5e-7), not the polynomial of the bf16 tier.
template <int ACT, int OUT_MODE, bool HAS_RES, typename CoordFn>
__device__ __forceinline__ void halo3_epilogue_hp(const ds_conv_params& p, f32x4 (&acc)[XT][WT], int b, int n0, int outHW, const float* shl,
                                                  CoordFn coord, float& s1, float& s2, float ga, int lane) {
    static_assert(OUT_MODE == 1 || OUT_MODE == 2, "1 = split bf16 planes, 2 = fp32");
    const int g = lane >> 4, n_loc = 8 * g;                  // channels 32 k + 8 g .. + 7 of group k (EPI_CH)
    constexpr unsigned ES = OUT_MODE == 2 ? 4u : 2u;                 // bytes per element of the out tensor as described by out_C
    const unsigned sample_bytes = (unsigned)outHW * p.out_C * ES;
    char* const obase = reinterpret_cast<char*>(p.out) + (size_t)b * sample_bytes;
    const char* const rbase = reinterpret_cast<const char*>(p.res) + (size_t)b * sample_bytes;
    const rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(obase, (short)0, (int)sample_bytes, 0x00020000);
    const rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(HAS_RES ? rbase : obase), (short)0, HAS_RES ? (int)sample_bytes : 0, 0x00020000);
    const int cout_v = (p.Cout + 7) / 8 * 8;
    const bool nine = (p.gn_ab != nullptr || p.gn_part != nullptr) && p.ncls == 9;
    const unsigned lo_off = (unsigned)p.Cout * 2u;                    // split planes: the lo plane starts Cout channels further
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        const ConvCoord c = coord(i);
        int cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
        if (!nine) cls = 0;
        const float* shrow = shl + (c.ok ? cls : 9) * BN + n_loc;
        const float gai = c.ok ? ga : 0.f;
        const unsigned o = (unsigned)(c.pix * p.out_C + p.out_c0 + n0 + n_loc) * ES;
        u32x4 rres[HAS_RES ? 6 : 1];
        if constexpr (HAS_RES) {                                      // fp32 residual: 3 x 8 channels = 6 x 16 B (tile k: channel EPI_CH(k) + 8 g)
            static_assert(!HAS_RES || OUT_MODE == 2, "a residual comes with the fp32 output mode");
#pragma unroll
            for (int k = 0; k < 6; ++k)
                rres[k] = buf_ld16(rs_r, rbase, (c.ok && n0 + n_loc + EPI_CH(k) < cout_v) ? o + (unsigned)EPI_CH(k) * 4u : VOFF_NONE, 0u, DS_BX_RES);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x4 sa = *reinterpret_cast<const f32x4*>(shrow + 32 * k), sb = *reinterpret_cast<const f32x4*>(shrow + 32 * k + 4);
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = act_const<ACT>(fmaf(gai, acc[i][2 * k][r], sa[r]));
                v[4 + r] = act_const<ACT>(fmaf(gai, acc[i][2 * k + 1][r], sb[r]));
            }
            if constexpr (HAS_RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += __uint_as_float(rres[2 * k][r]);
                    v[4 + r] += __uint_as_float(rres[2 * k + 1][r]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += v[e];
                s2 = fmaf(v[e], v[e], s2);
            }
            const bool okk = c.ok && n0 + n_loc + 32 * k < cout_v;
            if constexpr (OUT_MODE == 2) {
                buf_st16(rs_o, obase, okk ? o + 128u * k : VOFF_NONE, u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, DS_BX_OUT);
                buf_st16(rs_o, obase, okk ? o + 128u * k + 16u : VOFF_NONE, u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, DS_BX_OUT);
            } else {
                u32x4 hi8, lo8;
                ds_split8(v, hi8, lo8);
                buf_st16(rs_o, obase, okk ? o + 64u * k : VOFF_NONE, hi8, DS_BX_OUT);
                buf_st16(rs_o, obase, okk ? o + 64u * k + lo_off : VOFF_NONE, lo8, DS_BX_OUT);
            }
        }
    }
}

// ---- fp32 output (+ fp32 residual) with line-sized stores: the 16 x 96 fp32 values of one accumulator tile row go to a wave-private LDS tile
// ([pixel][96], pitch 400 B) and come back as 384 consecutive 16-byte pieces, 24 per pixel; the residual is loaded on the contiguous side.
// In the accumulator layout a store instruction is 64 separate 16-byte pieces: to_qkv at 256 x 64 wrote its 3.2 GB at 2.3 TB/s while a fill
// kernel writes 6.8 (tools/ubench/bw_probe.py) — the L2's request rate, not bytes.  Used by conv1x1_x3 and by the split-precision 3x3 (conv2).
constexpr int EPI_F32_PITCH = 400, EPI_F32_WAVE = 16 * EPI_F32_PITCH;
template <bool NCLS9, bool HAS_RES, typename CoordFn, typename Coord2Fn>
__device__ __forceinline__ void halo3_epilogue_rows_f32(const ds_conv_params& p, f32x4 (&acc)[XT][WT], int b, int n0, int outHW, const float* shl,
                                                        CoordFn coord, Coord2Fn coord2, char* stage, float& s1, float& s2, float ga, int lane) {
    const int m = lane & 15, g = lane >> 4, n_loc = 8 * g;
    const unsigned sample_bytes = (unsigned)outHW * p.out_C * 4u;
    char* const obase = reinterpret_cast<char*>(p.out) + (size_t)b * sample_bytes;
    const char* const rbase = reinterpret_cast<const char*>(p.res) + (size_t)b * sample_bytes;
    const rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(obase, (short)0, (int)sample_bytes, 0x00020000);
    const rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(HAS_RES ? rbase : obase), (short)0, HAS_RES ? (int)sample_bytes : 0, 0x00020000);
    const int cout_v = (p.Cout + 7) / 8 * 8;
    const bool nine = (p.gn_ab != nullptr || p.gn_part != nullptr) && p.ncls == 9;
    int pxl[6], pq[6];                                       // contiguous side: piece lane + 64 t of the tile row = pixel pxl, channels 4 pq .. 4 pq + 3
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int pc = lane + 64 * t;
        pxl[t] = pc / 24;
        pq[t] = pc - 24 * pxl[t];
    }
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        unsigned off[6];
        u32x4 rres[HAS_RES ? 6 : 1];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const ConvCoord c2 = coord2(i, pxl[t]);
            const unsigned bad = (unsigned)!c2.ok | (unsigned)(n0 + 4 * pq[t] >= cout_v);
            off[t] = (((unsigned)(c2.pix * p.out_C + p.out_c0 + n0 + 4 * pq[t]) * 4u) & 0x7fffffffu) | (bad << 31);
            if constexpr (HAS_RES) rres[t] = buf_ld16(rs_r, rbase, off[t], 0u, DS_BX_RES);
        }
        const ConvCoord c = coord(i);
        int cls = 0;
        if constexpr (NCLS9) cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
        if (!nine) cls = 0;
        const float* const shrow = shl + (c.ok ? cls : 9) * BN + n_loc;
        const float gi = c.ok ? ga : 0.f;
#pragma unroll
        for (int k = 0; k < WT; ++k) {
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shrow + EPI_CH(k));
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(gi, acc[i][k][r], sh[r]);
            *reinterpret_cast<f32x4*>(stage + m * EPI_F32_PITCH + (n_loc + EPI_CH(k)) * 4) = v;
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + pxl[t] * EPI_F32_PITCH + pq[t] * 16);
            if constexpr (HAS_RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += __uint_as_float(rres[t][r]);
            }
            // (pieces beyond the image / the valid channels hold zeros: lanes without a pixel multiply by 0 and read the table's zero row, padded
            // channels have zero weights and shifts, and a masked residual load returns 0 — they leave the statistics alone)
            s1a += v[0] + v[2];
            s1b += v[1] + v[3];
            s2a = fmaf(v[0], v[0], fmaf(v[2], v[2], s2a));
            s2b = fmaf(v[1], v[1], fmaf(v[3], v[3], s2b));
            buf_st16(rs_o, obase, off[t], __builtin_bit_cast(u32x4, v), DS_BX_OUT);
        }
    }
    s1 += s1a + s1b;
    s2 += s2a + s2b;
}

}  // namespace
