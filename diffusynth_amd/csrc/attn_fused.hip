// Fused linear attention for the U-Net's Residual(PreNorm(LinearCrossAttentionAdd)) block
// (diffusion_components.py:142-152,252-293), bf16, 4 heads x 32 (gfx950).
//
// The unfused path materialises the 384-channel qkv tensor (the largest tensor of the network) and reads
// it twice.  Here x is the only input stream:
//
//   pass 1 (context):  per wave = one head; k,v tiles = x[32 px] . Wk^T, Wv^T on MFMA (PreNorm folded into the
//                      weights + a per-channel shift), two sweeps over the segment (max, then exp/accumulate).
//                      The 32x32 accumulators of P = exp(k - max) and V already have the operand layout of the
//                      next product (column on the lane, rows in registers), so ctx += P^T V is two more MFMAs
//                      on the bf16-packed accumulators — no LDS, no transposes.
//   combine:           (linattn.hip) merges the segments' (max, sum, ctx).
//   pass 2 (output):   per wave = one 32-pixel tile; q^T = Wq . x^T (pixels on lanes => the softmax over d runs
//                      over registers), Y_h = ctx_h^T . q~_h and Z = Wout . [Y_0..Y_3] are chained the same way
//                      (accumulator tile as the B operand; the A operands ctx^T / Wout use the matching permuted
//                      k order), + bias, GroupNorm partials, row-major store through a wave-private LDS stage.
//
// HBM traffic per pixel: 2 reads of x (second one L2-resident) + 1 read of x + 1 write of y,
// instead of x + 2.67x(384 ch) + ... of the unfused chain.
#include "common.hpp"
#include "conv_epilogue.hpp"   // permlane32_swap

#ifndef DS_ATTN_ABL
#define DS_ATTN_ABL 0   // diagnostic builds only: bit0 no y stores, bit1 no statistics, bit2 no Z phase, bit3 no q softmax, bit4 no x prefetch
#endif
int ds_linattn_launch_combine(const ds_attn_params* p, hipStream_t st);  // linattn.hip

namespace {

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
    return o;
}
__device__ __forceinline__ int acc_row(int r, int fh) { return (r & 3) + 8 * (r >> 2) + 4 * fh; }

constexpr int PARTF = 32 + 32 + 1024;
constexpr float LOG2E = 1.44269504088896340736f;
__device__ __forceinline__ float exp2f_fast(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32 (flushes denormal results)

// Both passes share one structure: a block is 4 waves = the 4 heads; it walks a contiguous range of 32*T-pixel
// groups of one sample.  The group's x rows (contiguous in NHWC) are fetched with fully coalesced 16-byte loads one
// group ahead, staged in LDS (rows padded by 16 B: an odd number of 16-byte slots, conflict-free ds_read_b128) and
// shared by the four heads.  Every weight fragment a wave needs lives in its registers for the whole kernel.
template <int NKS, int T>
struct XStage {
    static constexpr int C = NKS * 16, TP = 32 * T, RS = 2 * C + 16;   // row stride in bytes
    static constexpr int BYTES = TP * RS;
    static constexpr int PIECES = TP * 2 * NKS;                        // 16-byte pieces of one group
    static constexpr int IT = (PIECES + 255) / 256;
    u32x4 r[IT];
    // unconditional loads (clamped address + select): a load under a branch would serialise the prefetch
    __device__ __forceinline__ void load(const bf16* x, int N, int group) {
        const long base = (long)group * TP * C;
        const long lim = (long)N * C;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int piece = threadIdx.x + it * 256;
            const long e = base + (long)piece * 8;
            const bool ok = piece < PIECES && e < lim;
            const u32x4 v = DS_LD(u32x4, x + (ok ? e : 0), DS_BX_SRC0);
            r[it] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* buf) const {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int piece = threadIdx.x + it * 256;
            if (PIECES % 256 == 0 || piece < PIECES) {
                const int row = piece / (2 * NKS), col = piece - row * (2 * NKS);
                *reinterpret_cast<u32x4*>(buf + row * RS + col * 16) = r[it];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ pass 1
// k, v tiles of this head: acc layout = (lane: d resp. e, registers: 16 pixel rows).  Online softmax over the pixels
// (running max per d, shared by the two lane halves), ctx^T[e][d] += V^T P with the accumulators themselves as the
// MFMA operands: the rescale by exp(m_old - m_new) is then a per-LANE factor.
template <int NKS, int T>
__global__ __launch_bounds__(256, NKS >= 24 ? 1 : 2) void attn_fused_ctx_kernel(const ds_attn_fused_params p) {
    using XS = XStage<NKS, T>;
    constexpr int C = XS::C, TP = XS::TP, RS = XS::RS;
    constexpr int KCH = NKS < 12 ? NKS : 12;                    // fragment reads in flight per chunk
    extern __shared__ __attribute__((aligned(16))) char sm[];   // x[2][XS::BYTES]
    const int seg = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, head = threadIdx.x >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int ngroups = (p.N + TP - 1) / TP, per = (ngroups + p.nseg - 1) / p.nseg;
    const int g0 = seg * per, g1 = min(ngroups, g0 + per);
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    const int nk = 128 + head * 32 + frow, nv = 256 + head * 32 + frow;
    float* out = p.part + (((size_t)b * 4 + head) * p.nseg + seg) * PARTF;

    float m = -INFINITY, ls = 0.f;
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    if (g0 < g1) {                                    // block-uniform
        XS xs;
        xs.load(x, p.N, g0);
        bf16x8 Wk[NKS], Wv[NKS];
        {
            const bf16* wk = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)nk * C + fh * 8;
            const bf16* wv = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)nv * C + fh * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                Wk[ks] = DS_LD(bf16x8, wk + ks * 16, DS_BX_W);
                Wv[ks] = DS_LD(bf16x8, wv + ks * 16, DS_BX_W);
            }
        }
        float ga, gam;
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
        else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
        const float shk = DS_LD(float, p.t1 + nk, DS_BX_T1) - gam * DS_LD(float, p.t2 + nk, DS_BX_T2);
        const float shv = DS_LD(float, p.t1 + nv, DS_BX_T1) - gam * DS_LD(float, p.t2 + nv, DS_BX_T2);
        const float ga2 = ga * LOG2E, shk2 = shk * LOG2E;
        xs.store(sm);
        xs.load(x, p.N, g0 + 1 < g1 ? g0 + 1 : g0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int cur = (g - g0) & 1;
            const char* xb = sm + cur * XS::BYTES + frow * RS + fh * 16;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x16 ak, av;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ak[r] = 0.f;
                    av[r] = 0.f;
                }
                // all fragment reads of a chunk are issued before its first MFMA (left alone, the scheduler reuses one
                // register quad and waits out the full LDS latency in front of every MFMA)
#pragma unroll
                for (int k0 = 0; k0 < NKS; k0 += KCH) {
                    bf16x8 xf[KCH];
#pragma unroll
                    for (int ks = 0; ks < KCH; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xb + t * 32 * RS + (k0 + ks) * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KCH; ++ks) {
                        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], Wk[k0 + ks], ak, 0, 0, 0);
                        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], Wv[k0 + ks], av, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // softmax in the log2 domain: ga2 / shk2 carry log2(e), so every exponential is a bare v_exp_f32
                const int px0 = g * TP + t * 32;
                const bool full = px0 + 32 <= p.N;                    // wave-uniform: only the last group can be ragged
                // (ga2 = rstd * log2 e > 0: the maximum of the affine image is the affine image of the raw maximum, and the softmax
                // argument ga2 * ak + shk2 - max folds into one fma per element)
                float mr = -INFINITY;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mr = fmaxf(mr, ak[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (px0 + acc_row(r, fh) >= p.N) ak[r] = -INFINITY;     // exp2(-inf) = 0
                        mr = fmaxf(mr, ak[r]);
                    }
                }
                mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
                const float mt = fmaf(ga2, mr, shk2);
                const float mn = fmaxf(m, mt);                       // finite: every group holds >= 1 real pixel
                const float sc = exp2f_fast(m - mn);                 // m = -inf on the first tile -> 0
                m = mn;
                const float cexp = shk2 - mn;
                float P[16], V[16], psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    P[r] = exp2f_fast(fmaf(ga2, ak[r], cexp));
                    V[r] = ga * av[r] + shv;
                    psum += P[r];
                }
                ls = ls * sc + psum;
                if (__any(sc != 1.0f)) {                             // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) ctx[r] *= sc;
                }
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(V), pack8(P), ctx, 0, 0, 0);          // ctx^T[e][d]
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(V + 8), pack8(P + 8), ctx, 0, 0, 0);
            }
            xs.store(sm + (cur ^ 1) * XS::BYTES);          // group g+1 (loaded a full iteration ago)
            xs.load(x, p.N, g + 2 < g1 ? g + 2 : g);       // group g+2 stays in flight across the next iteration
            __syncthreads();
        }
        ls += __shfl_xor(ls, 32, 64);
    }
    if (fh == 0) {
        DS_ST(float, out + frow, DS_BX_AUX0, m * (1.0f / LOG2E));   // back to the natural-log domain of the combine kernel; lane = d
        DS_ST(float, out + 32 + frow, DS_BX_AUX0, ls);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) DS_ST(float, out + 64 + frow * 32 + acc_row(r, fh), DS_BX_AUX0, ctx[r]);   // ctx[d][e]: d on the lane, e in registers
}

// ------------------------------------------------------------------------------------------------ pass 2
// wave = head: q^T = Wq_h . x^T (pixels on lanes => softmax over d runs over registers), Y_h = ctx_h^T . q~_h with the
// accumulator tile as the B operand; the four heads' Y tiles (bf16, already B-operand shaped) are exchanged through LDS
// and each wave then produces the 32-channel blocks {wave, wave+4, ..} of Z = Wout . [Y_0..Y_3] + bias.  Z's accumulators
// hold one pixel per lane: v_permlane32_swap pairs the lane halves into 8 consecutive channels = one 16-byte store.
// blocks per sample: the whole grid is resident at once (2 blocks per CU; 1 for the 384-channel variant) — a second,
// partial round of blocks would double the kernel's duration
static inline int out_blocks(int ngroups, int B, int C) {
    static const int cap96 = getenv("DS_ATTN_CAP") ? atoi(getenv("DS_ATTN_CAP")) : 512;
    int nb = (C == 384 ? 256 : (C == 96 ? cap96 : 512)) / B;
    if (nb > ngroups) nb = ngroups;
    if (nb < 1) nb = 1;
    const int per = (ngroups + nb - 1) / nb;
    return (ngroups + per - 1) / per;
}

template <int NKS, int T>
__global__ __launch_bounds__(256, NKS >= 24 ? 1 : 2) void attn_fused_out_kernel(const ds_attn_fused_params p) {
    using XS = XStage<NKS, T>;
    constexpr int C = XS::C, CB = C / 32, TP = XS::TP, RS = XS::RS;
    constexpr int NCB = (CB + 3) / 4;
    constexpr int KCH = NKS < 12 ? NKS : 12;                    // fragment reads in flight per chunk
    constexpr int YBYTES = 4 * T * 2 * 1024;
    extern __shared__ __attribute__((aligned(16))) char sm[];   // x[2][XS::BYTES] | y[2][4 heads][T][2][64 lanes] x 16 B
    __shared__ __attribute__((aligned(16))) float red[8];
    __shared__ __attribute__((aligned(16))) float sbias[C];     // to_out bias: loop-invariant, but a global load inside the loop is re-issued per tile
    const int b = blockIdx.y, lane = threadIdx.x & 63, head = threadIdx.x >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    char* const ybase = sm + 2 * XS::BYTES;
    const int ngroups = (p.N + TP - 1) / TP, per = (ngroups + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per, g1 = min(ngroups, g0 + per);
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)b * p.N * C;
    float s1 = 0.f, s2 = 0.f;
    if (g0 < g1) {                                    // block-uniform
        XS xs;
        xs.load(x, p.N, g0);
        // ---- register-resident operands of this wave
        bf16x8 Wq[NKS], Wo[NCB][8], cA[2];
        {
            const bf16* wq = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)(head * 32 + frow) * C + fh * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) Wq[ks] = DS_LD(bf16x8, wq + ks * 16, DS_BX_W);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int cb = head + 4 * c < CB ? head + 4 * c : 0;
                const bf16* wo = reinterpret_cast<const bf16*>(p.wout_perm) + (size_t)(cb * 32 + frow) * 128 + fh * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) Wo[c][k] = DS_LD(bf16x8, wo + k * 16, DS_BX_AUX1);
            }
            const float* ctx = p.ctx + ((size_t)b * 4 + head) * 1024;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float ca[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ca[j] = DS_LD(float, ctx + (16 * s + 8 * (j >> 2) + 4 * fh + (j & 3)) * 32 + frow, DS_BX_AUX2);
                cA[s] = pack8(ca);
            }
        }
        float ga, gam;
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
        else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
        float shq[16];                                   // additive part of q (fold shift + label_q) for this lane's 16 rows d
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = head * 32 + acc_row(r, fh);
            shq[r] = LOG2E * (DS_LD(float, p.t1 + d, DS_BX_T1) - gam * DS_LD(float, p.t2 + d, DS_BX_T2) +
                              (p.label_q ? DS_LD(float, p.label_q + (size_t)b * p.lq_stride + d, DS_BX_AUX3) : 0.f));
        }
        const float ga2 = ga * LOG2E;                    // softmax over d in the log2 domain: bare v_exp_f32
        xs.store(sm);
        for (int i = threadIdx.x; i < C; i += 256) sbias[i] = DS_LD(float, p.bias_out + i, DS_BX_BIAS);
        xs.load(x, p.N, g0 + 1 < g1 ? g0 + 1 : g0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int cur = (g - g0) & 1;
            const char* xb = sm + cur * XS::BYTES + frow * RS + fh * 16;
            bf16x8* const yw = reinterpret_cast<bf16x8*>(ybase + cur * YBYTES) + (head * T * 2) * 64 + lane;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x16 aq;
#pragma unroll
                for (int r = 0; r < 16; ++r) aq[r] = 0.f;
#pragma unroll
                for (int k0 = 0; k0 < NKS; k0 += KCH) {
                    bf16x8 xf[KCH];
#pragma unroll
                    for (int ks = 0; ks < KCH; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xb + t * 32 * RS + (k0 + ks) * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KCH; ++ks) aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wq[k0 + ks], xf[ks], aq, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float q[16], mxq = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    q[r] = ga2 * aq[r] + shq[r];
                    mxq = fmaxf(mxq, q[r]);
                }
                if constexpr (!(DS_ATTN_ABL & 8)) {
                mxq = fmaxf(mxq, __shfl_xor(mxq, 32, 64));
                float sq = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    q[r] = exp2f_fast(q[r] - mxq);
                    sq += q[r];
                }
                sq += __shfl_xor(sq, 32, 64);
                const float inv = p.scale / sq;
#pragma unroll
                for (int r = 0; r < 16; ++r) q[r] *= inv;
                }
                // Y_h[e][px] = sum_d ctx[d][e] q~[d][px]: A = ctx^T in the permuted k order of the accumulator operand
                f32x16 Y;
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[r] = 0.f;
                Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cA[0], pack8(q), Y, 0, 0, 0);
                Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cA[1], pack8(q + 8), Y, 0, 0, 0);
                float yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = Y[r];
                yw[(t * 2 + 0) * 64] = pack8(yv);
                yw[(t * 2 + 1) * 64] = pack8(yv + 8);
            }
            xs.store(sm + (cur ^ 1) * XS::BYTES);          // group g+1 (loaded a full iteration ago)
            if constexpr (!(DS_ATTN_ABL & 16)) xs.load(x, p.N, g + 2 < g1 ? g + 2 : g);   // group g+2 stays in flight across the Z phase and the next Q phase
            __syncthreads();
            // ---- Z[c][px] = sum_{h,e} Wout[c][h*32+e] Y_h[e][px] + bias[c] for this wave's channel blocks
            const bf16x8* const yr = reinterpret_cast<const bf16x8*>(ybase + cur * YBYTES) + lane;
#pragma unroll
            for (int t = 0; t < ((DS_ATTN_ABL & 4) ? 0 : T); ++t) {
                bf16x8 yB[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) yB[k] = yr[(((k >> 1) * T + t) * 2 + (k & 1)) * 64];
                __builtin_amdgcn_sched_barrier(0);
                const int px = g * TP + t * 32 + frow;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const int cb = head + 4 * c;
                    if (cb < CB) {                   // wave-uniform
                        f32x16 Z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) Z[r] = 0.f;
#pragma unroll
                        for (int k = 0; k < 8; ++k) Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wo[c][k], yB[k], Z, 0, 0, 0);
                        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(Z));   // MFMA write -> inline-asm read (tied to Z: cannot move off the accumulators)
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            float v[8];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[k] = Z[8 * q2 + k];
                                v[4 + k] = Z[8 * q2 + 4 + k];
                                permlane32_swap(v[k], v[4 + k]);
                            }
                            const int c0 = cb * 32 + 16 * q2 + 8 * fh;
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + c0), b1 = *reinterpret_cast<const f32x4*>(sbias + c0 + 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[k] += b0[k];
                                v[4 + k] += b1[k];
                            }
                            if (px < p.N) {
                                if constexpr (!(DS_ATTN_ABL & 1)) vec16_store<bf16>(yout + (size_t)px * C + c0, v, DS_BX_OUT);
                                else if (v[0] == 12345.678f) yout[0] = (bf16)v[1];
#pragma unroll
                                for (int k = 0; k < ((DS_ATTN_ABL & 2) ? 0 : 8); ++k) {
                                    s1 += v[k];
                                    s2 += v[k] * v[k];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

__global__ void pack_attn_kernel(const float* wqkv, const float* gamma, const float* wout, bf16* wq_out, bf16* wo_out, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 384 * C) wq_out[i] = (bf16)(wqkv[i] * gamma[i % C]);
    if (i < C * 128) {
        // column permutation inside each head block of 32: position s*16 + h*8 + j  <-  e = 16s + 8(j>>2) + 4h + (j&3)
        const int c = i / 128, pcol = i % 128, hh = pcol / 32, pp = pcol % 32, s = pp / 16, h = (pp / 8) & 1, j = pp & 7;
        const int e = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        wo_out[i] = (bf16)wout[c * 128 + hh * 32 + e];
    }
}

int check(const ds_attn_fused_params* p) {
    DS_REQUIRE(p && p->x && p->wqkv && p->t1 && p->t2 && (p->gn_ab || p->gn_part) && p->part && p->ctx, "attn_fused: null pointer");
    DS_REQUIRE(p->C == 96 || p->C == 192 || p->C == 384, "attn_fused: C=%d unsupported (96, 192, 384)", p->C);
    DS_REQUIRE(p->B > 0 && p->N > 0 && p->nseg > 0, "attn_fused: bad sizes");
    if (!ds_aligned16(p->x) || !ds_aligned16(p->wqkv)) DS_FAIL(DS_EALIGN, "attn_fused: pointers must be 16-byte aligned");
    return DS_OK;
}

}  // namespace

extern "C" int ds_pack_attn_fused(const float* wqkv, const float* gamma, const float* wout, void* wq_bf16, void* wo_perm_bf16, int C,
                                  void* stream) {
    DS_REQUIRE(wqkv && gamma && wout && wq_bf16 && wo_perm_bf16 && C > 0, "pack_attn_fused: bad args");
    hipLaunchKernelGGL(pack_attn_kernel, dim3((384 * C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wqkv, gamma,
                       wout, (bf16*)wq_bf16, (bf16*)wo_perm_bf16, C);
    DS_CHECK_LAUNCH("pack_attn_fused");
    return DS_OK;
}

namespace {

#if DS_BOUNDS
void attn_publish_bounds(const ds_attn_fused_params* p, int kernel, int stats_parts, hipStream_t st) {
    DsBxHost h(kernel);
    h.set(DS_BX_SRC0, p->x, (long long)p->B * p->N * p->C * 2);
    h.set(DS_BX_W, p->wqkv, (long long)384 * p->C * 2);
    h.set(DS_BX_AUX1, p->wout_perm, (long long)p->C * 128 * 2);
    h.set(DS_BX_T1, p->t1, 384 * 4).set(DS_BX_T2, p->t2, 384 * 4);
    h.set(DS_BX_GNAB, p->gn_ab, (long long)p->B * 2 * 4);
    h.set(DS_BX_GNPART, p->gn_part, (long long)p->B * p->gn_parts * 2 * 4);
    h.set(DS_BX_AUX0, p->part, (long long)p->B * 4 * p->nseg * PARTF * 4);
    h.set(DS_BX_AUX2, p->ctx, (long long)p->B * 4 * 1024 * 4);
    h.set(DS_BX_AUX3, p->label_q, p->label_q ? ((long long)(p->B - 1) * p->lq_stride + 128) * 4 : 0);
    h.set(DS_BX_BIAS, p->bias_out, (long long)p->C * 4);
    h.set(DS_BX_OUT, p->y, (long long)p->B * p->N * p->C * 2);
    h.set(DS_BX_STATS, p->stats_part, (long long)p->B * stats_parts * 2 * 4);
    h.set(DS_BX_RES, p->mfold, p->mfold ? (long long)p->B * p->C * 256 : 0);
    h.publish(st);
}
#endif

template <int NKS, int T>
int launch_ctx(const ds_attn_fused_params* p, hipStream_t st) {
    auto kern = attn_fused_ctx_kernel<NKS, T>;
    constexpr int lds = 2 * XStage<NKS, T>::BYTES;
    DS_SET_MAX_LDS(kern, lds, "attn_fused_ctx");
#if DS_BOUNDS
    attn_publish_bounds(p, DS_K_ATTN_CTX, 0, st);
#endif
    hipLaunchKernelGGL(kern, dim3(p->nseg, p->B), dim3(256), lds, st, *p);
    DS_CHECK_LAUNCH("attn_fused_ctx");
    return DS_OK;
}

template <int NKS, int T>
int launch_out(const ds_attn_fused_params* p, hipStream_t st) {
    auto kern = attn_fused_out_kernel<NKS, T>;
    constexpr int lds = 2 * XStage<NKS, T>::BYTES + 2 * 4 * T * 2 * 1024;
    DS_SET_MAX_LDS(kern, lds, "attn_fused_out");
    const int ngroups = (p->N + 32 * T - 1) / (32 * T);
#if DS_BOUNDS
    attn_publish_bounds(p, DS_K_ATTN_OUT, out_blocks(ngroups, p->B, p->C), st);
#endif
    hipLaunchKernelGGL(kern, dim3(out_blocks(ngroups, p->B, p->C), p->B), dim3(256), lds, st, *p);
    DS_CHECK_LAUNCH("attn_fused_out");
    return DS_OK;
}

}  // namespace
#include "attn_out2.hpp"
namespace {

// Which generation runs (ds_attn_fused_params.gen; 0 = by batch): the second-generation kernels stage 50 - 100 KB of weights per block and
// walk pixel tiles with them — measured per level at U-Net batch 16 / 32 / 64 / 128 (tools/ab.sh -m attn): context pass 53 / 70 / 95 / 176 us
// against 32 / 52 / 92 / 222 us of the first generation at C = 96, output pass 42 / 62 / 129 / 233 against 35 / 67 / 151 / 272
static inline bool ctx2_exists(int C, int N) { return C == 96 || C == 192 || (C == 384 && N >= 1024); }
static inline bool use_ctx2(const ds_attn_fused_params* p) {
    static const bool off = getenv("DS_ATTN_V1") != nullptr || getenv("DS_ATTN_CTX1") != nullptr;      // A/B switches
    static const bool no384 = getenv("DS_ATTN_CTX2_NO384") != nullptr;
    if (off || p->gen == 1 || !ctx2_exists(p->C, p->N) || (p->C == 384 && no384)) return false;
    return p->gen == 2 || p->B >= 96;
}
static inline bool use_out2(const ds_attn_fused_params* p) {
    static const bool off = getenv("DS_ATTN_V1") != nullptr;
    static const bool only96 = getenv("DS_ATTN_OUT2_96") != nullptr;      // A/B switch: first-generation output pass at C = 192
    if (off || p->gen == 1 || !p->mfold || !(p->C == 96 || (p->C == 192 && !only96))) return false;
    return p->gen == 2 || p->B >= (p->C == 96 ? 32 : 96);
}

// pixels per group: 64 where the image is large enough to keep every CU busy with fewer, longer iterations
static inline int group_t(int C, int N) { static const int f = getenv("DS_ATTN_T1") ? 1 : 0; return (C == 96 && N >= 4096 && !f) ? 2 : 1; }

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_attn_fused(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" int ds_attn_fused_context(const ds_attn_fused_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int T = group_t(p->C, p->N);
    if (use_ctx2(p)) {
#if DS_BOUNDS
        attn_publish_bounds(p, DS_K_ATTN_CTX, 0, st);
#endif
        rc = attn_ctx2_launch(p, st);
    } else if (p->C == 96) rc = T == 2 ? launch_ctx<6, 2>(p, st) : launch_ctx<6, 1>(p, st);
    else if (p->C == 192) rc = launch_ctx<12, 1>(p, st);
    else rc = launch_ctx<24, 1>(p, st);
    if (rc) return rc;
    ds_attn_params q;
    memset(&q, 0, sizeof(q));
    q.B = p->B; q.N = p->N; q.heads = 4; q.nseg = p->nseg; q.part = p->part; q.ctx = p->ctx;
    return ds_linattn_launch_combine(&q, st);
}

extern "C" int ds_attn_fused_output(const ds_attn_fused_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    DS_REQUIRE(p->wout_perm && p->bias_out && p->y, "attn_fused_output: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (use_out2(p)) {
        DS_REQUIRE(ds_aligned16(p->mfold) && ds_aligned16(p->y), "attn_fused_output: mfold / y must be 16-byte aligned");
#if DS_BOUNDS
        attn_publish_bounds(p, DS_K_ATTN_OUT, attn_out2_blocks(p->N, p->B, p->C), st);
#endif
        return attn_out2_launch(p, st);
    }
    const int T = group_t(p->C, p->N);
    if (p->C == 96) return T == 2 ? launch_out<6, 2>(p, st) : launch_out<6, 1>(p, st);
    if (p->C == 192) return launch_out<12, 1>(p, st);
    return launch_out<24, 1>(p, st);
}

extern "C" int ds_attn_fused_segments_gen(int B, int N, int C, int gen) {
    ds_attn_fused_params q;
    memset(&q, 0, sizeof(q));
    q.B = B; q.N = N; q.C = C; q.gen = gen;
    const int ntiles = (N + 31) / 32;
    if (use_ctx2(&q)) {
        int s = (C == 384 ? 1024 : 2048) / (B > 0 ? B : 1);
        if (s > 64) s = 64;
        if (s > ntiles) s = ntiles;
        return s < 1 ? 1 : s;
    }
    int s = N / 128;                                            // >= 128-pixel segments; <= 32 keeps the combine short
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

extern "C" int ds_attn_fused_segments(int B, int N, int C) { return ds_attn_fused_segments_gen(B, N, C, 0); }

extern "C" int ds_attn_fused_stats_parts(const ds_attn_fused_params* p) {
    if (use_out2(p)) return attn_out2_blocks(p->N, p->B, p->C);
    const int tp = 32 * group_t(p->C, p->N);
    return out_blocks((p->N + tp - 1) / tp, p->B, p->C);
}
