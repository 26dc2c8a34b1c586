// Output pass of the fused linear attention block, second generation (bf16, C = 96 / 192; gfx950).
// Residual(PreNorm(LinearCrossAttentionAdd)): diffusion_components.py:142-152,252-293 — q projection, softmax over d, ctx^T q,
// to_out 1x1 + bias, GroupNorm partials of the result.
//
// What changed against attn_fused_out_kernel (attn_fused.hip), and why (r02 ablations of that kernel at 256 x 64, U-Net batch 128:
// 42 % of its time in the to_out phase, 17 % in the q softmax, 16 % in 32-byte y stores; one of its four waves idle in the to_out
// phase at C = 96; a block barrier and an LDS exchange between the two phases of every tile):
//
//   * wave = pixel tile, not head.  A wave computes q for all four heads of its own 32 pixels (pixels on lanes: the softmax over d
//     runs over registers), keeps the softmaxed q~ as MFMA B operands in registers and multiplies them straight into the output
//     channels — no LDS exchange, no barrier inside the loop, no idle wave, and the waves of a SIMD are in different phases by
//     construction (MFMA of one beside the exponentials of another).
//   * to_out is folded into the context once per sample (attn_fold_out_kernel): Z = Wout . (ctx^T q~) = (Wout . ctx^T) . q~ =: M_b . q~,
//     M_b [C][128] per sample — the Y = ctx^T q~ product and its bf16 rounding disappear (2 of 14 MFMAs per head-tile at C = 96).
//   * the rows of M_b are stored in the order that makes accumulator register r of lane half fh the channel 16 fh + r of its 32-channel
//     block: a lane holds 16 CONSECUTIVE channels of its pixel = two 16-byte stores, without v_permlane32_swap (whose inline-asm
//     form needed hand-counted s_nop padding behind the MFMA) and without an LDS transpose.
//   * Wq (128 x C) and M_b (C x 128) live in LDS (53 KB at C = 96: three 4-wave blocks per CU; 104 KB at C = 192: one 8-wave block),
//     read as A fragments; x fragments come straight from global memory (a lane reads 16 bytes of its pixel row per K step: the rows'
//     other bytes are the next K steps' — served by L1) and are prefetched one tile ahead into the registers the q MFMAs just freed.
// (included by attn_fused.hip: one translation unit, one bounds table)
#pragma once

#include <type_traits>
namespace {

__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ int acc_row32(int r, int fh) { return (r & 3) + 8 * (r >> 2) + 4 * fh; }   // row of register r in a 32x32 accumulator
__device__ __forceinline__ bf16x8 pack8f(const float* v) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
    return o;
}

template <int NKS>
struct O2 {
    static constexpr int C = 16 * NKS, CB = C / 32;
    static constexpr int WQ_RS = 2 * C + 16;            // LDS row strides: an odd number of 16-byte slots (conflict-free ds_read_b128 over 16 rows)
    static constexpr int M_RS = 2 * 128 + 16;
    static constexpr int OFF_WQ = 0, OFF_M = 128 * WQ_RS, OFF_SHQ = OFF_M + C * M_RS, OFF_BIAS = OFF_SHQ + 128 * 4, OFF_RED = OFF_BIAS + C * 4;
    static constexpr int LDS = OFF_RED + 64;
};

// M_b[c][h*32 + d] = sum_e Wout[c][h*32 + e] ctx[b][h][d][e], bf16, in the operand layout of attn_out2_kernel:
//   row  m of channel block cb  <-  channel cb*32 + 16*((m>>2)&1) + (m&3) + 4*(m>>3)      (accumulator register r of lane half fh = channel 16 fh + r)
//   column h*32 + s*16 + kg*8 + j  <-  d = 16 s + 8 (j>>2) + 4 kg + (j&3)                   (the k order of a packed 32x32 accumulator used as B operand)
// wo_perm is ds_pack_attn_fused's copy of Wout (columns of a head in that same k order: only the pairing of e with ctx's e matters here).
__global__ __launch_bounds__(256) void attn_fold_out_kernel(const float* ctx, const bf16* wo_perm, bf16* mfold, int C) {
    __shared__ float sctx[4 * 32 * 33];
    const int cb = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) sctx[(i >> 5) * 33 + (i & 31)] = ctx[(size_t)b * 4096 + i];     // [h*32 + d][e]
    __syncthreads();
    for (int o = tid; o < 32 * 128; o += 256) {
        const int m = o >> 7, pos = o & 127;
        const int c = cb * 32 + 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
        const int h = pos >> 5, pp = pos & 31, s = pp >> 4, kg = (pp >> 3) & 1, j = pp & 7;
        const int d = 16 * s + 8 * (j >> 2) + 4 * kg + (j & 3);
        const bf16* w = wo_perm + (size_t)c * 128 + h * 32;
        const float* cr = sctx + (h * 32 + d) * 33;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int s2 = q >> 4, k2 = (q >> 3) & 1, j2 = q & 7;
            acc += (float)w[q] * cr[16 * s2 + 8 * (j2 >> 2) + 4 * k2 + (j2 & 3)];
        }
        mfold[((size_t)b * C + cb * 32 + m) * 128 + pos] = (bf16)acc;
    }
}

template <int NKS, int NW>
__global__ __launch_bounds__(NW * 64, NKS == 6 ? 3 : 2) void attn_out2_kernel(const ds_attn_fused_params p, const bf16* mfold, const int tiles_per_block) {
    using G = O2<NKS>;
    constexpr int C = G::C, CB = G::CB, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    float* const shq = reinterpret_cast<float*>(sm + G::OFF_SHQ);
    float* const sbias = reinterpret_cast<float*>(sm + G::OFF_BIAS);
    float* const red = reinterpret_cast<float*>(sm + G::OFF_RED);
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)b * p.N * C;
    const int ntiles = (p.N + 31) >> 5;
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);

    // x fragments of a tile: lane (pixel n, k group kg) reads channels ks*16 + kg*8 .. + 7 of its pixel for every K step
    bf16x8 xf[NKS];
    auto load_x = [&](int t) {
        // (pixels past the end of a ragged last tile read pixel 0 instead: their columns are computed and never stored)
        const int px = t * 32 + n;
        const bf16* row = x + (size_t)(px < p.N ? px : 0) * C + kg * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xf[ks] = DS_LD(bf16x8, row + ks * 16, DS_BX_SRC0);
    };
    if (t0 + wave < t1) load_x(t0 + wave);

    // ---- block prologue: Wq rows 0..127 of the packed qkv weights and this sample's folded to_out matrix -> LDS
    {
        // (all loads of the block's operands requested before the first LDS write: as a `for (i = tid; ...; i += NT)` loop every iteration
        // was load -> s_waitcnt vmcnt(0) -> ds_write, twelve serial memory round trips per block)
        const char* wq = reinterpret_cast<const char*>(p.wqkv);
        const char* mb = reinterpret_cast<const char*>(mfold) + (size_t)b * C * 256;
        constexpr int WIT = 128 * 2 * NKS / NT, MIT = C * 16 / NT;
        static_assert(WIT * NT == 128 * 2 * NKS && MIT * NT == C * 16, "whole staging iterations");
        u32x4 wst[WIT], mst[MIT];
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, row = i / (2 * NKS), col = i - row * (2 * NKS);
            wst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wq + ((size_t)row * C * 2 + col * 16)), DS_BX_W);
        }
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, row = i >> 4, col = i & 15;
            mst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(mb + ((size_t)row * 256 + col * 16)), DS_BX_RES);
        }
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, row = i / (2 * NKS), col = i - row * (2 * NKS);
            *reinterpret_cast<u32x4*>(sm + G::OFF_WQ + row * G::WQ_RS + col * 16) = wst[k];
        }
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, row = i >> 4, col = i & 15;
            *reinterpret_cast<u32x4*>(sm + G::OFF_M + row * G::M_RS + col * 16) = mst[k];
        }
        float ga, gam;
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
        else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
        // additive part of q in the log2 domain, in accumulator order: entry (h, fh, r) = row d = h*32 + acc_row32(r, fh)
        for (int i = tid; i < 128; i += NT) {
            const int d = (i >> 5) * 32 + acc_row32(i & 15, (i >> 4) & 1);
            shq[i] = LOG2E * (DS_LD(float, p.t1 + d, DS_BX_T1) - gam * DS_LD(float, p.t2 + d, DS_BX_T2) +
                              (p.label_q ? DS_LD(float, p.label_q + (size_t)b * p.lq_stride + d, DS_BX_AUX3) : 0.f));
        }
        for (int i = tid; i < C; i += NT) sbias[i] = DS_LD(float, p.bias_out + i, DS_BX_BIAS);
        if (tid == 0) red[15] = ga * LOG2E;
    }
    __syncthreads();
    const float ga2 = red[15];
    __syncthreads();                       // (red is reused by the statistics reduction at the end)

    const char* const wq_l = sm + G::OFF_WQ + n * G::WQ_RS + kg * 16;      // A fragment (head h, K step ks): + h*32*WQ_RS + ks*32
    const char* const m_l = sm + G::OFF_M + n * G::M_RS + kg * 16;         // A fragment (block cb, step hs):  + cb*32*M_RS + hs*32
    float s1 = 0.f, s2 = 0.f;
    for (int t = t0 + wave; t < t1; t += NW) {
        // ---- q^T_h = Wq_h . x^T for the four heads (pixels on lanes, 16 rows d per lane half), softmax over d, scale, pack as B operands
        bf16x8 qB[4][2];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x16 aq;
            // (scheduling fences between the heads: left alone, the scheduler hoists every fragment read of the tile to its top and spills)
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 wf[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wq_l + h * 32 * G::WQ_RS + ks * 32);
                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // literal zero accumulator: no register clearing
                aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[0], z, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < NKS; ++ks) aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], aq, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (h == 3 && t + NW < t1) load_x(t + NW);               // the x fragments are consumed: fetch the next tile's into the same registers
            float q[16], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shq + (h * 2 + kg) * 16 + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[4 * k + e] = fmaf(ga2, aq[4 * k + e], sh[e]);
                    mx = fmaxf(mx, q[4 * k + e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                q[r] = exp2_hw(q[r] - mx);
                sq += q[r];
            }
            sq += __shfl_xor(sq, 32, 64);
            const float inv = p.scale / sq;
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] *= inv;
            qB[h][0] = pack8f(q);
            qB[h][1] = pack8f(q + 8);
        }
        // ---- Z[c][px] = sum_{h,d} M_b[c][h*32 + d] q~_h[d][px] + bias[c]: lane = pixel, registers = 16 consecutive channels
        const int px = t * 32 + n;
        const bool okp = px < p.N;
        bf16* const yrow = yout + (size_t)(okp ? px : 0) * C + 16 * kg;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            // The kernel is bound by its VALU instruction count (profiles/r03_attn_pmc.txt): the accumulators START from the bias (four
            // LDS reads instead of 16 clears + 16 adds), and the GroupNorm statistics of the block's output are summed from the packed
            // bf16 values that are stored — what the consumer normalises — two channels per v_dot2c_f32_bf16.
            f32x16 Z;
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 mf[8];
#pragma unroll
                for (int hs = 0; hs < 8; ++hs) mf[hs] = *reinterpret_cast<const bf16x8*>(m_l + cb * 32 * G::M_RS + hs * 32);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + cb * 32 + 16 * kg + 4 * k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Z[4 * k + e] = bv[e];
                }
#pragma unroll
                for (int hs = 0; hs < 8; ++hs) Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf[hs], qB[hs >> 1][hs & 1], Z, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = Z[r];
            if (okp) {
                const bf16x8 y0 = pack8f(v), y1 = pack8f(v + 8);
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(yrow + cb * 32), DS_BX_OUT, y0);
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(yrow + cb * 32 + 8), DS_BX_OUT, y1);
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x2_t a = {y0[2 * j], y0[2 * j + 1]}, c = {y1[2 * j], y1[2 * j + 1]};
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(a, one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2_f32_bf16(a, a, s2, false);
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(c, one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2_f32_bf16(c, c, s2, false);
                }
            }
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------ context pass, second generation
// The same structure for pass 1 (k / v projection, softmax over the pixels, ctx += P^T V): wave = 32-pixel tile x all four heads (C = 96) or two of them (C = 192).
// Wk and Wv (rows 128..383 of the packed qkv weights) live in LDS and are read as B fragments, x fragments come straight from global memory
// (one tile ahead), every wave keeps its own running maximum / sum / 32 x 32 context per head in registers and writes them as ONE segment of
// the (max, sum, ctx) partials that attn_ctx_combine merges — no LDS staging of x, no block barrier per pixel group, no exchange.
template <int NKS>
struct C2 {
    static constexpr int C = 16 * NKS, RS = 2 * C + 16;
    static constexpr int LDS = 256 * RS;                 // 53 KB (C = 96: two 4-wave blocks per CU), 102 KB (C = 192: one 8-wave block)
};

template <int NKS, int NW, int HPW, int HG>
__global__ __launch_bounds__(NW * 64, 2) void attn_ctx2_kernel(const ds_attn_fused_params p) {
    using G = C2<NKS>;
    // HG head groups over blockIdx.z (C = 384: the k / v weights of all four heads do not fit in LDS, a block takes two and x is read twice);
    // a block's HB heads are split over wave groups of HPW heads; segments per block: a wave owns HPW heads of one segment
    constexpr int HB = 4 / HG, NSB = NW * HPW / HB;
    constexpr int C = G::C, NT = NW * 64, KCH = 3;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    // segment = wave: p.nseg is the caller's (ds_attn_fused_segments_gen: one round of blocks, 2048 / B segments — the bf16 tier's partial
    // sums therefore depend on the batch a sample travels in, to fp32 rounding of the combine: DESIGN §3)
    const int ntiles = (p.N + 31) >> 5, per = (ntiles + p.nseg - 1) / p.nseg;
    const int nseg = p.nseg, seg = blockIdx.x * NSB + wave % NSB, hl0 = (wave / NSB) * HPW, h0 = blockIdx.z * HB + hl0;   // local / absolute first head
    const int t0 = min(ntiles, seg * per), t1 = min(ntiles, t0 + per);     // an empty segment writes the neutral partial (max = -inf, sum = 0)

    bf16x8 xf[NKS];
    auto load_x = [&](int t) {
        // (pixels past the end of a ragged last tile read pixel 0 instead: finite values whose softmax weight is set to exactly 0)
        const int px = t * 32 + n;
        const bf16* row = x + (size_t)(px < p.N ? px : 0) * C + kg * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xf[ks] = DS_LD(bf16x8, row + ks * 16, DS_BX_SRC0);
    };
    if (t0 < t1) load_x(t0);
    {
        // rows 128 .. 255 of the packed qkv weights: k heads, 256 .. 383: v heads; LDS: this block's HB k heads, then its HB v heads
        // (all requested before the first LDS write, in two halves: see attn_out2_kernel)
        const char* wkv = reinterpret_cast<const char*>(p.wqkv);
        constexpr int WIT = 2 * HB * 32 * 2 * NKS / NT;
        static_assert(WIT * NT == 2 * HB * 32 * 2 * NKS && WIT % 2 == 0, "whole staging iterations");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 wst[WIT / 2];
#pragma unroll
            for (int k = 0; k < WIT / 2; ++k) {
                const int i = tid + (half * (WIT / 2) + k) * NT, row = i / (2 * NKS), col = i - row * (2 * NKS);
                const int src = row < HB * 32 ? 128 + blockIdx.z * HB * 32 + row : 256 + blockIdx.z * HB * 32 + (row - HB * 32);
                wst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wkv + ((size_t)src * C * 2 + col * 16)), DS_BX_W);
            }
#pragma unroll
            for (int k = 0; k < WIT / 2; ++k) {
                const int i = tid + (half * (WIT / 2) + k) * NT, row = i / (2 * NKS), col = i - row * (2 * NKS);
                *reinterpret_cast<u32x4*>(sm + row * G::RS + col * 16) = wst[k];
            }
        }
    }
    float ga, gam;
    if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
    else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
    const float ga2 = ga * LOG2E;
    float shk2[HPW], m[HPW], ls[HPW];
    f32x16 ctx[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int nk = 128 + (h0 + h) * 32 + n;                       // this lane's column d of head h0 + h
        shk2[h] = LOG2E * (DS_LD(float, p.t1 + nk, DS_BX_T1) - gam * DS_LD(float, p.t2 + nk, DS_BX_T2));
        m[h] = -INFINITY;
        ls[h] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx[h][r] = 0.f;
    }
    __syncthreads();
    const char* const wl = sm + (hl0 * 32 + n) * G::RS + kg * 16;    // B fragment of this wave's head h, K step ks: + h*32*RS + ks*32 (Wv: + HB*32 rows)
    // The weight fragments are read one chunk (KCH K-steps of one head's Wk and Wv) AHEAD of the MFMAs that use them, across heads and tiles
    // (they do not depend on the tile): the LDS latency of a chunk hides behind the previous chunk's MFMAs resp. the previous head's softmax.
    constexpr int NCH = NKS / KCH;
    bf16x8 wk[2][KCH], wv[2][KCH];
    auto read_w = [&](bf16x8 (&dk)[KCH], bf16x8 (&dv)[KCH], int h, int k0) {
#pragma unroll
        for (int ks = 0; ks < KCH; ++ks) {
            dk[ks] = *reinterpret_cast<const bf16x8*>(wl + h * 32 * G::RS + (k0 + ks) * 32);
            dv[ks] = *reinterpret_cast<const bf16x8*>(wl + (HB + h) * 32 * G::RS + (k0 + ks) * 32);       // Wv rows: HB heads further on
        }
    };
    read_w(wk[0], wv[0], 0, 0);
    auto tile = [&](const int t, auto ragged) {
        const int px0 = t * 32;
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            f32x16 ak, av;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cur = (h * NCH + c) & 1, nx = (h * NCH + c + 1) % (HPW * NCH);
                __builtin_amdgcn_sched_barrier(0);
                read_w(wk[cur ^ 1], wv[cur ^ 1], nx / NCH, (nx % NCH) * KCH);
#pragma unroll
                for (int ks = 0; ks < KCH; ++ks) {
                    if (c == 0 && ks == 0) {                          // accumulators start from the literal zero (no register clearing)
                        const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0], wk[cur][0], z, 0, 0, 0);     // rows = pixels, columns = d
                        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0], wv[cur][0], z, 0, 0, 0);
                    } else {
                        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[c * KCH + ks], wk[cur][ks], ak, 0, 0, 0);
                        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[c * KCH + ks], wv[cur][ks], av, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (h == HPW - 1 && t + 1 < t1) load_x(t + 1);           // the x fragments are consumed: the next tile's go into the same registers
            // online softmax over the pixels in the log2 domain (ga2 > 0: the maximum is taken on the raw accumulators)
            if constexpr (decltype(ragged)::value) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (px0 + acc_row32(r, kg) >= p.N) ak[r] = -INFINITY;       // exp2(-inf) = 0
            }
            float mr = ak[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mr = fmaxf(mr, ak[r]);
            mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
            const float mn = fmaxf(m[h], fmaf(ga2, mr, shk2[h]));     // finite: a tile holds >= 1 real pixel
            const float sc = exp2_hw(m[h] - mn);                      // m = -inf on the wave's first tile -> 0
            m[h] = mn;
            const float cexp = shk2[h] - mn;
            // This kernel is bound by its VALU instruction count (profiles/r03_attn_pmc.txt), so per element only what must be: the exponential's
            // argument (1 fma), the exponential, one add for the denominator, the two bf16 conversions.  v enters the context RAW — its
            // normalisation is affine and is applied to the finished context (ctx = ga * ctx_raw + shv[e] * sum_px P, see the write-out).
            float P[16], V[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                P[r] = exp2_hw(fmaf(ga2, ak[r], cexp));
                V[r] = av[r];
            }
            const bf16x8 p0 = pack8f(P), p1 = pack8f(P + 8);
            float psum = 0.f, ps1 = 0.f;                              // (two chains; v_dot2c_f32_bf16 on the packed pairs measured slower:
#pragma unroll
            for (int r = 0; r < 8; ++r) {                             //  21 cycles per instruction beside a busy matrix pipe against 2 x 8.4)
                psum += P[r];
                ps1 += P[8 + r];
            }
            psum += ps1;
            ls[h] = fmaf(ls[h], sc, psum);
            if (__any(sc != 1.0f)) {                                  // the running maximum rarely moves after the first tiles
#pragma unroll
                for (int r = 0; r < 16; ++r) ctx[h][r] *= sc;
            }
            ctx[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8f(V), p0, ctx[h], 0, 0, 0);          // ctx^T[e][d]
            ctx[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8f(V + 8), p1, ctx[h], 0, 0, 0);
        }
    };
    {
        // only the last tile of a sample can be ragged (wave-uniform): it runs the masking copy of the tile body, every other tile the plain one
        const int t1f = (t1 == ntiles && (p.N & 31)) ? t1 - 1 : t1;
        for (int t = t0; t < t1f; ++t) tile(t, std::false_type{});
        if (t1f < t1 && t1f >= t0) tile(t1f, std::true_type{});
    }
    // ---- this wave's segment of the partials: [32 max (natural log domain)][32 sum][ctx[d][e]] per head
    if (seg >= nseg) return;
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        float* out = p.part + (((size_t)b * 4 + h0 + h) * nseg + seg) * (32 + 32 + 1024);
        const float lsum = ls[h] + __shfl_xor(ls[h], 32, 64);
        if (kg == 0) {
            DS_ST(float, out + n, DS_BX_AUX0, m[h] * (1.0f / LOG2E));
            DS_ST(float, out + 32 + n, DS_BX_AUX0, lsum);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = acc_row32(r, kg), nv = 256 + (h0 + h) * 32 + e;                 // v's normalisation, applied once: + shv[e] * sum_px P[px][d]
            const float shv = DS_LD(float, p.t1 + nv, DS_BX_T1) - gam * DS_LD(float, p.t2 + nv, DS_BX_T2);
            DS_ST(float, out + 64 + n * 32 + e, DS_BX_AUX0, fmaf(ga, ctx[h][r], shv * lsum));
        }
    }
}

}  // namespace

static int attn_ctx2_launch(const ds_attn_fused_params* p, hipStream_t st) {
    if (p->C == 96) {
        auto kern = attn_ctx2_kernel<6, 4, 4, 1>;
        DS_SET_MAX_LDS(kern, C2<6>::LDS, "attn_ctx2");
        hipLaunchKernelGGL(kern, dim3((p->nseg + 3) / 4, p->B), dim3(256), C2<6>::LDS, st, *p);
    } else if (p->C == 192) {
        auto kern = attn_ctx2_kernel<12, 8, 4, 1>;
        DS_SET_MAX_LDS(kern, C2<12>::LDS, "attn_ctx2");
        hipLaunchKernelGGL(kern, dim3((p->nseg + 7) / 8, p->B), dim3(512), C2<12>::LDS, st, *p);
    } else {
        auto kern = attn_ctx2_kernel<24, 8, 2, 2>;             // two heads per block: half of the k / v weights (100 KB)
        DS_SET_MAX_LDS(kern, C2<24>::LDS / 2, "attn_ctx2");
        hipLaunchKernelGGL(kern, dim3((p->nseg + 7) / 8, p->B, 2), dim3(512), C2<24>::LDS / 2, st, *p);
    }
    DS_CHECK_LAUNCH("attn_ctx2");
    return DS_OK;
}

// blocks per sample: every CU busy with few, long-lived blocks (a block pays 53 / 104 KB of operand staging)
static int attn_out2_blocks(int N, int B, int C) {
    const int ntiles = (N + 31) / 32;
    const int nw = C == 96 ? 4 : 8, per_cu = C == 96 ? 3 : 1;
    int nb = (256 * per_cu + B - 1) / B;                      // blocks per sample that fill the chip once
    const int max_nb = (ntiles + nw - 1) / nw;                // at least one tile per wave
    if (nb > max_nb) nb = max_nb;
    if (nb < 1) nb = 1;
    const int per = (ntiles + nb - 1) / nb;
    return (ntiles + per - 1) / per;
}

static int attn_out2_launch(const ds_attn_fused_params* p, hipStream_t st) {
    void* const mfold = p->mfold;
    const int C = p->C, nb = attn_out2_blocks(p->N, p->B, C);
    const int ntiles = (p->N + 31) / 32, per = (ntiles + nb - 1) / nb;
    hipLaunchKernelGGL(attn_fold_out_kernel, dim3(C / 32, p->B), dim3(256), 0, st, p->ctx, reinterpret_cast<const bf16*>(p->wout_perm),
                       reinterpret_cast<bf16*>(mfold), C);
    DS_CHECK_LAUNCH("attn_fold_out");
    if (C == 96) {
        auto kern = attn_out2_kernel<6, 4>;
        DS_SET_MAX_LDS(kern, O2<6>::LDS, "attn_out2");
        hipLaunchKernelGGL(kern, dim3(nb, p->B), dim3(256), O2<6>::LDS, st, *p, reinterpret_cast<const bf16*>(mfold), per);
    } else {
        auto kern = attn_out2_kernel<12, 8>;
        DS_SET_MAX_LDS(kern, O2<12>::LDS, "attn_out2");
        hipLaunchKernelGGL(kern, dim3(nb, p->B), dim3(512), O2<12>::LDS, st, *p, reinterpret_cast<const bf16*>(mfold), per);
    }
    DS_CHECK_LAUNCH("attn_out2");
    return DS_OK;
}
