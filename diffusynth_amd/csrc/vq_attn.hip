// Fused LinearAttention block of the VQGAN (VQGAN.py:246-272: one head of 32, softmax over n on k only), bf16 tier (gfx950).
//
//     qkv = to_qkv(x);  ctx[d][e] = sum_n softmax_n(k[d])[n] v[e][n];  out[e][n] = sum_d ctx[d][e] q[d][n];  y = to_out(out) + nin_shortcut(x)
//
// q enters LINEARLY (no softmax over d, no bias in to_qkv), so everything after the context is one 1x1 convolution with a PER-SAMPLE weight:
//     y = (Wnin + Wout ctx_b^T Wq) x + (b_out + b_nin) = W_b x + bias.
// The unfused chain (to_qkv on the generic 1x1 tile, context, output, the merged to_out | nin_shortcut 1x1: r03) moved a 96-channel qkv tensor
// and a 32-channel attention output through HBM: 819 us at 256 x 128 x 80, batch 64.  Here x is the only activation stream, read twice:
//   pass 1 (vq_attn_ctx):   block = 4 waves, each with its own 32-pixel tile of a 128-pixel group staged in LDS (coalesced 16-byte loads two
//                           groups ahead, rows padded to an odd number of 16-byte slots); k, v tiles on v_mfma_f32_32x32x16_bf16 with the weight
//                           fragments in registers, online softmax over the pixels in the log2 domain, ctx^T += V^T P with the accumulators as
//                           operands (attn_fused.hip, first generation); one (max, sum, ctx) partial per wave, merged by attn_ctx_combine;
//   fold    (vq_attn_fold): W_b per sample in fp32, stored as bf16 [roundup(C, 32)][C];
//   pass 2 (vq_attn_apply): W_b in LDS; wave = 32-pixel tile, its x fragments in registers, y^T = W_b x^T per 32-channel block, + bias, written over
//                           the wave's own x rows in LDS and stored as whole contiguous rows; the per-channel (sum, sum of squares) of the stored
//                           values go to the statistics slots the following Normalize finishes (ds_gn_stats_finish) — no statistics pass.
#include "common.hpp"

int ds_linattn_launch_combine(const ds_attn_params* p, hipStream_t st);   // linattn.hip

namespace {

constexpr int PARTF = 32 + 32 + 1024;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr int GP = 128;                                          // pixels per group: one 32-pixel tile per wave

__device__ __forceinline__ int acc_row(int r, int fh) { return (r & 3) + 8 * (r >> 2) + 4 * fh; }
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    unsigned h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = __builtin_bit_cast(unsigned, __builtin_convertvector((ds_f32x2{v[2 * j], v[2 * j + 1]}), ds_bf16x2));
    return __builtin_bit_cast(bf16x8, (u32x4{h[0], h[1], h[2], h[3]}));
}

// a group's x rows: 256 threads x NKS 16-byte pieces, unconditional loads (clamped address + select), see attn_fused.hip
template <int NKS>
struct VX {
    static constexpr int C = NKS * 16, RS = 2 * C + 16, BYTES = GP * RS, PPR = 2 * NKS;     // row stride in bytes, pieces per row
    u32x4 r[NKS];
    __device__ __forceinline__ void load(const bf16* x, int N, int group) {
        const long base = (long)group * GP * C, lim = (long)N * C;
#pragma unroll
        for (int it = 0; it < NKS; ++it) {
            const long e = base + (long)(threadIdx.x + it * 256) * 8;
            const bool ok = e < lim;
            const u32x4 v = DS_LD(u32x4, x + (ok ? e : 0), DS_BX_SRC0);
            r[it] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* buf) const {
#pragma unroll
        for (int it = 0; it < NKS; ++it) {
            const int piece = threadIdx.x + it * 256;
            const int row = piece / PPR, col = piece - row * PPR;
            *reinterpret_cast<u32x4*>(buf + row * RS + col * 16) = r[it];
        }
    }
};

// ------------------------------------------------------------------------------------------------ pass 1
template <int NKS>
__global__ __launch_bounds__(256, NKS > 5 ? 1 : 2) void vq_attn_ctx_kernel(const ds_vq_attn_params p) {
    using XS = VX<NKS>;
    constexpr int C = XS::C, RS = XS::RS;
    extern __shared__ __attribute__((aligned(16))) char sm[];    // x[2][XS::BYTES]
    const int blk = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int ngroups = (p.N + GP - 1) / GP, per = (ngroups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = blk * per, g1 = min(ngroups, g0 + per);
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    float* out = p.part + ((size_t)b * p.nseg + blk * 4 + wave) * PARTF;

    float m = -INFINITY, ls = 0.f;
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    if (g0 < g1) {                                               // block-uniform
        XS xs;
        xs.load(x, p.N, g0);
        bf16x8 Wk[NKS], Wv[NKS];
        {
            const bf16* wk = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)(32 + frow) * C + fh * 8;
            const bf16* wv = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)(64 + frow) * C + fh * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                Wk[ks] = DS_LD(bf16x8, wk + ks * 16, DS_BX_W);
                Wv[ks] = DS_LD(bf16x8, wv + ks * 16, DS_BX_W);
            }
        }
        xs.store(sm);
        xs.load(x, p.N, g0 + 1 < g1 ? g0 + 1 : g0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int cur = (g - g0) & 1;
            const char* xb = sm + cur * XS::BYTES + (wave * 32 + frow) * RS + fh * 16;
            const int px0 = g * GP + wave * 32;
            if (px0 < p.N) {                                     // wave-uniform: a tile beyond a ragged image has nothing to add
                f32x16 ak, av;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ak[r] = 0.f;
                    av[r] = 0.f;
                }
                bf16x8 xf[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xb + ks * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], Wk[ks], ak, 0, 0, 0);     // [pixel][d]: d on the lane, pixels in registers
                    av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], Wv[ks], av, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                float mr = -INFINITY;
                if (px0 + 32 <= p.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mr = fmaxf(mr, ak[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (px0 + acc_row(r, fh) >= p.N) ak[r] = -INFINITY;                        // exp2(-inf) = 0
                        mr = fmaxf(mr, ak[r]);
                    }
                }
                mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
                const float mn = fmaxf(m, mr * LOG2E);           // finite: the tile holds >= 1 real pixel
                const float sc = __builtin_amdgcn_exp2f(m - mn); // m = -inf on the first tile -> 0
                m = mn;
                float P[16], V[16], psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    P[r] = __builtin_amdgcn_exp2f(fmaf(LOG2E, ak[r], -mn));
                    V[r] = av[r];
                    psum += P[r];
                }
                ls = ls * sc + psum;
                if (__any(sc != 1.0f)) {                         // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) ctx[r] *= sc;
                }
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(V), pack8(P), ctx, 0, 0, 0);            // ctx^T[e][d]
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(V + 8), pack8(P + 8), ctx, 0, 0, 0);
            }
            xs.store(sm + (cur ^ 1) * XS::BYTES);                // group g+1 (loaded a full iteration ago)
            xs.load(x, p.N, g + 2 < g1 ? g + 2 : g);             // group g+2 stays in flight across the next iteration
            __syncthreads();
        }
        ls += __shfl_xor(ls, 32, 64);
    }
    if (fh == 0) {
        DS_ST(float, out + frow, DS_BX_AUX0, m * (1.0f / LOG2E));        // natural-log domain of the combine kernel; lane = d
        DS_ST(float, out + 32 + frow, DS_BX_AUX0, ls);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) DS_ST(float, out + 64 + frow * 32 + acc_row(r, fh), DS_BX_AUX0, ctx[r]);   // ctx[d][e]: d on the lane, e in registers
}

// ------------------------------------------------------------------------------------------------ fold
// W_b[co][ci] = Wnin[co][ci] + sum_e Wout[co][e] T[e][ci],  T[e][ci] = sum_d ctx_b[d][e] Wq[d][ci]   (fp32; bf16 on store; rows >= C zero).
// Block = (sample, 16 output rows); every block recomputes the 32 x C matrix T (5 k fma per thread at most) rather than waiting for a
// kernel of its own — as one block per sample this step took 60 us (3 200 dependent global loads per thread), more than the context pass.
constexpr int FOLD_ROWS = 16;
__global__ __launch_bounds__(256) void vq_attn_fold_kernel(const ds_vq_attn_params p) {
    extern __shared__ __attribute__((aligned(16))) float fs[];  // ctx[1024] | Wout rows [16][32] | Wq[32][C] | T[32][C]
    const int b = blockIdx.x, r0 = blockIdx.y * FOLD_ROWS, tid = threadIdx.x, C = p.C, CP = (C + 31) / 32 * 32;
    float* const cx = fs;
    float* const wo = fs + 1024;
    float* const wqs = fs + 1024 + FOLD_ROWS * 32;
    float* const T = wqs + 32 * C;
    // every global operand in ONE round trip (independent loads; a loop of load -> use iterations paid one L2 latency each: 36 - 72 us)
    for (int i = tid; i < 1024; i += 256) cx[i] = DS_LD(float, p.ctx + (size_t)b * 1024 + i, DS_BX_AUX2);
    for (int i = tid; i < FOLD_ROWS * 32; i += 256) wo[i] = r0 + i / 32 < C ? DS_LD(float, p.wout + (size_t)r0 * 32 + i, DS_BX_AUX1) : 0.f;
    for (int i = tid; i < 32 * C; i += 256) wqs[i] = DS_LD(float, p.wq + i, DS_BX_T1);
    __syncthreads();
    const int CQ = C >> 2;                                       // column quads (C % 16 == 0)
    {
        // T: thread = (e, one of 8 quad groups), its ctx column in registers; one ds_read_b128 of Wq per four fma (a ds_read_b32 per fma made
        // this kernel LDS-issue-bound: 30 - 50 us)
        const int e = tid & 31, cg = tid >> 5;
        float cc[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) cc[d] = cx[d * 32 + e];
        for (int cq = cg; cq < CQ; cq += 8) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wqs + d * C + 4 * cq);
                a[0] = fmaf(cc[d], w4[0], a[0]); a[1] = fmaf(cc[d], w4[1], a[1]); a[2] = fmaf(cc[d], w4[2], a[2]); a[3] = fmaf(cc[d], w4[3], a[3]);
            }
            *reinterpret_cast<f32x4*>(T + e * C + 4 * cq) = a;
        }
    }
    __syncthreads();
    {
        // W_b rows r0 .. r0 + 15: thread = (row, one of 16 quad groups), its Wout row in registers
        const int rr = tid >> 4, qg = tid & 15, co = r0 + rr;
        float wr[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) wr[e] = wo[rr * 32 + e];
        bf16* const wf = reinterpret_cast<bf16*>(p.wfold) + ((size_t)b * CP + co) * C;
        for (int cq = qg; cq < CQ; cq += 16) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (co < C) {
                if (p.wnin) a = DS_LD(f32x4, p.wnin + (size_t)co * C + 4 * cq, DS_BX_T2);
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(T + e * C + 4 * cq);
                    a[0] = fmaf(wr[e], t4[0], a[0]); a[1] = fmaf(wr[e], t4[1], a[1]); a[2] = fmaf(wr[e], t4[2], a[2]); a[3] = fmaf(wr[e], t4[3], a[3]);
                }
            }
            uint2 pk;
            pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector((ds_f32x2{a[0], a[1]}), ds_bf16x2));
            pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector((ds_f32x2{a[2], a[3]}), ds_bf16x2));
            DS_ST(uint2, wf + 4 * cq, DS_BX_RES, pk);
        }
    }
}

// ------------------------------------------------------------------------------------------------ pass 2
template <int NKS>
__global__ __launch_bounds__(256, NKS > 5 ? 1 : 2) void vq_attn_apply_kernel(const ds_vq_attn_params p) {
    using XS = VX<NKS>;
    constexpr int C = XS::C, RS = XS::RS, NCB = (C + 31) / 32, CP = NCB * 32, NP8 = C / 8, PPI = 64 / NP8, SIT = (32 + PPI - 1) / PPI;
    extern __shared__ __attribute__((aligned(16))) char sm[];    // x[2][XS::BYTES] | W_b[CP][RS] | bias[CP] fp32
    char* const wl = sm + 2 * XS::BYTES;
    float* const bl = reinterpret_cast<float*>(wl + CP * RS);
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int ngroups = (p.N + GP - 1) / GP, per = (ngroups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = blk * per, g1 = min(ngroups, g0 + per);
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    bf16* y = reinterpret_cast<bf16*>(p.y) + (size_t)b * p.N * C;
    // store-out map: lane -> (pixel of the iteration, 16-byte channel piece): the piece is the same in every iteration (per-channel statistics
    // in 16 registers), PPI whole rows = one contiguous run per instruction
    const int sp = lane / NP8, so = lane - sp * NP8;
    const bool s_act = lane < PPI * NP8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (g0 < g1) {                                               // block-uniform
        XS xs;
        xs.load(x, p.N, g0);
        {
            const bf16* wf = reinterpret_cast<const bf16*>(p.wfold) + (size_t)b * CP * C;
            for (int i = tid; i < CP * NP8; i += 256) {
                const int row = i / NP8, col = i - row * NP8;
                *reinterpret_cast<u32x4*>(wl + row * RS + col * 16) = DS_LD(u32x4, wf + (size_t)row * C + col * 8, DS_BX_RES);
            }
            for (int i = tid; i < CP; i += 256) bl[i] = i < C ? DS_LD(float, p.bias + i, DS_BX_BIAS) : 0.f;
        }
        xs.store(sm);
        xs.load(x, p.N, g0 + 1 < g1 ? g0 + 1 : g0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int cur = (g - g0) & 1;
            char* const tile = sm + cur * XS::BYTES + wave * 32 * RS;
            const int px0 = g * GP + wave * 32;
            if (px0 < p.N) {                                     // wave-uniform
                bf16x8 xf[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(tile + frow * RS + fh * 16 + ks * 32);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    f32x16 acc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + cb * 32 + 8 * j + 4 * fh);
                        acc[4 * j] = b4[0]; acc[4 * j + 1] = b4[1]; acc[4 * j + 2] = b4[2]; acc[4 * j + 3] = b4[3];
                    }
                    bf16x8 wfr[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) wfr[ks] = *reinterpret_cast<const bf16x8*>(wl + (cb * 32 + frow) * RS + ks * 32 + fh * 16);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[ks], xf[ks], acc, 0, 0, 0);   // [co][pixel]: pixel on the lane
                    // over the wave's own x rows (its fragments are in registers; the LDS operations of a wave complete in order)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c0 = cb * 32 + 8 * j + 4 * fh;
                        if (c0 < C) {                            // (compile-time for every (cb, j) except the half-filled last block)
                            uint2 pk;
                            pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector((ds_f32x2{acc[4 * j], acc[4 * j + 1]}), ds_bf16x2));
                            pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector((ds_f32x2{acc[4 * j + 2], acc[4 * j + 3]}), ds_bf16x2));
                            *reinterpret_cast<uint2*>(tile + frow * RS + c0 * 2) = pk;
                        }
                    }
                }
                if (s_act) {
#pragma unroll
                    for (int it = 0; it < SIT; ++it) {
                        const int pix = it * PPI + sp;
                        if (pix < 32 && px0 + pix < p.N) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(tile + pix * RS + so * 16);
                            DS_ST(u32x4, y + ((size_t)(px0 + pix) * C + so * 8), DS_BX_OUT, v);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                                s1[2 * e] += lo; s2[2 * e] = fmaf(lo, lo, s2[2 * e]);
                                s1[2 * e + 1] += hi; s2[2 * e + 1] = fmaf(hi, hi, s2[2 * e + 1]);
                            }
                        }
                    }
                }
            }
            xs.store(sm + (cur ^ 1) * XS::BYTES);
            xs.load(x, p.N, g + 2 < g1 ? g + 2 : g);
            __syncthreads();
        }
    }
    if (p.stats_ws) {
        // lanes of one channel piece (PPI per wave, 4 waves) -> one (sum, sum of squares) per channel and block
        float* const st = reinterpret_cast<float*>(sm);          // [256][16]
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            st[tid * 16 + e] = s1[e];
            st[tid * 16 + 8 + e] = s2[e];
        }
        __syncthreads();
        if (tid < C) {
            const int oct = tid >> 3, e = tid & 7;
            float a = 0.f, q = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int s = 0; s < PPI; ++s) {
                    const float* r = st + (w * 64 + s * NP8 + oct) * 16;
                    a += r[e];
                    q += r[8 + e];
                }
            float* ws = p.stats_ws + (((size_t)b * gridDim.x + blk) * C + tid) * 2;
            DS_ST(float, ws, DS_BX_STATS, a);
            DS_ST(float, ws + 1, DS_BX_STATS, q);
        }
    }
}

int check(const ds_vq_attn_params* p) {
    DS_REQUIRE(p && p->x && p->wqkv && p->part && p->ctx, "vq_attn: null pointer");
    DS_REQUIRE(p->C == 80 || p->C == 160, "vq_attn: C=%d unsupported (80, 160)", p->C);
    DS_REQUIRE(p->B > 0 && p->N > 0 && p->nseg >= 4 && p->nseg % 4 == 0, "vq_attn: bad sizes (nseg = ds_vq_attn_segments(B, N, C))");
    if (!ds_aligned16(p->x) || !ds_aligned16(p->wqkv)) DS_FAIL(DS_EALIGN, "vq_attn: pointers must be 16-byte aligned");
    return DS_OK;
}

#if DS_BOUNDS
void vq_publish_bounds(const ds_vq_attn_params* p, int kernel, hipStream_t st) {
    const int CP = (p->C + 31) / 32 * 32;
    DsBxHost h(kernel);
    h.set(DS_BX_SRC0, p->x, (long long)p->B * p->N * p->C * 2);
    h.set(DS_BX_W, p->wqkv, (long long)96 * p->C * 2);
    h.set(DS_BX_T1, p->wq, (long long)32 * p->C * 4).set(DS_BX_T2, p->wnin, (long long)p->C * p->C * 4);
    h.set(DS_BX_AUX1, p->wout, (long long)p->C * 32 * 4);
    h.set(DS_BX_AUX0, p->part, (long long)p->B * p->nseg * PARTF * 4);
    h.set(DS_BX_AUX2, p->ctx, (long long)p->B * 1024 * 4);
    h.set(DS_BX_RES, p->wfold, (long long)p->B * CP * p->C * 2);
    h.set(DS_BX_BIAS, p->bias, (long long)p->C * 4);
    h.set(DS_BX_OUT, p->y, (long long)p->B * p->N * p->C * 2);
    h.set(DS_BX_STATS, p->stats_ws, (long long)p->B * (p->nseg / 4) * p->C * 2 * 4);
    h.publish(st);
}
#endif

template <int NKS>
int launch_ctx(const ds_vq_attn_params* p, hipStream_t st) {
    auto kern = vq_attn_ctx_kernel<NKS>;
    constexpr int lds = 2 * VX<NKS>::BYTES;
    DS_SET_MAX_LDS(kern, lds, "vq_attn_ctx");
#if DS_BOUNDS
    vq_publish_bounds(p, DS_K_VQ_ATTN_CTX, st);
#endif
    hipLaunchKernelGGL(kern, dim3(p->nseg / 4, p->B), dim3(256), lds, st, *p);
    DS_CHECK_LAUNCH("vq_attn_ctx");
    return DS_OK;
}

template <int NKS>
int launch_apply(const ds_vq_attn_params* p, hipStream_t st) {
    auto kern = vq_attn_apply_kernel<NKS>;
    constexpr int C = NKS * 16, CP = (C + 31) / 32 * 32;
    constexpr int lds = 2 * VX<NKS>::BYTES + CP * VX<NKS>::RS + CP * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    DS_SET_MAX_LDS(kern, lds, "vq_attn_apply");
#if DS_BOUNDS
    vq_publish_bounds(p, DS_K_VQ_ATTN_APPLY, st);
#endif
    hipLaunchKernelGGL(kern, dim3(p->nseg / 4, p->B), dim3(256), lds, st, *p);
    DS_CHECK_LAUNCH("vq_attn_apply");
    return DS_OK;
}

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_vq_attn(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

// blocks per sample x 4 (one partial per wave): two resident rounds of 256-thread blocks (one at C = 160: 140 KB of LDS per block)
extern "C" int ds_vq_attn_segments(int B, int N, int C) {
    const int ngroups = (N + GP - 1) / GP;
    int nblk = (C <= 80 ? 512 : 256) / (B > 0 ? B : 1);
    if (nblk > 64) nblk = 64;
    if (nblk > ngroups) nblk = ngroups;
    if (nblk < 1) nblk = 1;
    return 4 * nblk;
}

extern "C" size_t ds_vq_attn_wfold_bytes(int B, int C) { return (size_t)B * ((C + 31) / 32 * 32) * C * 2; }

extern "C" int ds_vq_attn_context(const ds_vq_attn_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    rc = p->C == 80 ? launch_ctx<5>(p, st) : launch_ctx<10>(p, st);
    if (rc) return rc;
    ds_attn_params q;
    memset(&q, 0, sizeof(q));
    q.B = p->B; q.N = p->N; q.heads = 1; q.nseg = p->nseg; q.part = p->part; q.ctx = p->ctx;
    return ds_linattn_launch_combine(&q, st);
}

extern "C" int ds_vq_attn_output(const ds_vq_attn_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    DS_REQUIRE(p->wq && p->wout && p->bias && p->wfold && p->y, "vq_attn_output: null pointer");
    if (!ds_aligned16(p->wfold) || !ds_aligned16(p->y)) DS_FAIL(DS_EALIGN, "vq_attn_output: wfold / y must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    vq_publish_bounds(p, DS_K_VQ_ATTN_APPLY, st);
#endif
    hipLaunchKernelGGL(vq_attn_fold_kernel, dim3(p->B, ((p->C + 31) / 32 * 32) / FOLD_ROWS), dim3(256), (1024 + FOLD_ROWS * 32 + 64 * p->C) * 4, st, *p);
    DS_CHECK_LAUNCH("vq_attn_fold");
    return p->C == 80 ? launch_apply<5>(p, st) : launch_apply<10>(p, st);
}
