// Fused linear attention of the split-precision tier ("bf16x3"): Residual(PreNorm(LinearCrossAttentionAdd)) up to the output GroupNorm
// (diffusion_components.py:142-152,252-293) on fp32 tensors, every dense product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the bf16 matrix
// cores with fp32 accumulation (gfx950).
//
// The unfused chain of this tier wrote the 384-channel fp32 qkv tensor (3.2 GB per 256x64 block at U-Net batch 128) and read it twice:
// to_qkv (conv1x1_x3) + ds_linattn_context + ds_linattn_output + to_out were 17 ms of a 102 ms step.  Here x is the only fp32 input stream:
//
//   pass 1 (attn_x3_pass1):  wave = 32-pixel tile x the block's HB heads.  x arrives in 32-channel chunks: 16-byte pieces, 8 lanes per
//                            pixel = whole 128-byte lines, one chunk ahead in registers; split into hi / lo bf16 on its way to a
//                            wave-private LDS tile (no block barrier: LDS operations of a wave complete in order), read back as MFMA
//                            operands.  k, v and q of the block's heads are accumulated over the chunks from the hi / lo weight planes in
//                            LDS.  k, v: online softmax over the pixels, ctx += P^T V in split precision (the accumulators are the operands
//                            of the next product), one (max, sum, ctx) partial per wave, merged by attn_ctx_combine.  q: softmax over d in
//                            registers, scaled, split, and stored as ready-made B operands ("q planes": 16 fragments of 1 KB per tile).
//   fold (attn_x3_fold):     M_b = Wout . ctx_b^T per sample in fp32, split, in the A-operand layout of pass 2.
//   pass 2 (attn_x3_z):      Z = M_b . q~ + bias for 96 output channels per block from the q planes (B fragments straight from global
//                            memory: 1 KB contiguous per instruction), fp32 result through a wave-private LDS transpose as whole 128-byte
//                            lines, GroupNorm partials of the result.
//
// HBM traffic per pixel at C = 96: 384 B (x) + 512 B (q planes, written) + 512 B (read) + 384 B (y), against 384 + 3 x 1536 + 512 + 512 + 384
// of the unfused chain.  Where the weights of all four heads do not fit LDS the heads are split over blockIdx.z (x then comes from L2 for
// the other groups), and at C = 384 the k / v and the q projections run as two launches of pass 1.
#include "common.hpp"

int ds_linattn_launch_combine(const ds_attn_params* p, hipStream_t st);  // linattn.hip

namespace {

constexpr float LOG2E = 1.44269504088896340736f;
constexpr int PARTF = 32 + 32 + 1024;
constexpr int XS = 144;                  // row of a wave's staging tile: 64 B hi | 64 B lo | 16 B pad (nine 16-byte slots: conflict-free ds_read_b128)
constexpr int XTILE = 32 * XS;           // 4608 B per wave (also the 32 px x 32 ch fp32 transpose tile of pass 2: 128 + 16 B rows)
constexpr int NW = 8, NT = NW * 64;
constexpr int QFRAG = 1024, QTILE = 16 * QFRAG;     // q planes: [tile][plane (hi, lo)][head][s][lane half][pixel] x 16 B

__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ int acc_row32(int r, int fh) { return (r & 3) + 8 * (r >> 2) + 4 * fh; }   // row of register r in a 32x32 accumulator

// v[0..7] -> hi = bf16(v), lo = bf16(v - hi): the two MFMA operands of one fp32 operand (ds_split2: common.hpp)
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
    ds_split8(v, h, l);
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// a += x . w in split precision, small terms first
__device__ __forceinline__ f32x16 mma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// softmax over d of one head's q accumulator (rows d in registers, column = pixel on the lane; the two lane halves hold 16 rows each), scaled,
// split: registers 0..7 / 8..15 of lane (pixel, half) are the eight k values of K step s = 0 / 1 of the next product's B operand.
// sh = the additive part of q (fold + label) in the log2 domain for this lane half, in accumulator order.
__device__ __forceinline__ void q_softmax_split(const f32x16& aq, const float* sh, float ga2, float scale, bf16x8& qh0, bf16x8& ql0, bf16x8& qh1, bf16x8& ql1) {
    float q[16], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sh + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[4 * k + e] = fmaf(ga2, aq[4 * k + e], s4[e]);
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
    const float inv = scale / sq;
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] *= inv;
    split8(q, qh0, ql0);
    split8(q + 8, qh1, ql1);
}

// ---- x of one wave: 32-pixel tiles in 32-channel chunks.  A chunk = 32 px x 32 ch fp32 = 256 pieces of 16 B, four per lane: piece i*64 + lane
// = pixel i*8 + lane/8, column lane%8 (eight lanes cover one pixel's 128-byte line).  TWO chunks are in flight per wave (8 KB; 64 KB per CU):
// with one, the kernel ran at what 4 KB per wave buy against ~2 us of loaded HBM latency — 2.4 TB/s of reads.  stage() splits the oldest
// chunk into hi / lo bf16 on its way to the wave's LDS tile (row = pixel: 64 B hi | 64 B lo | 16 B pad), from which the MFMA operands of both
// products (x as A: k / v, x as B: q) are the same 16-byte reads.
template <int C>
struct XStream {
    f32x4 raw[2][4];
    const float* x;
    int N, spx, scol;
    char* xs;
    __device__ __forceinline__ void init(const float* x_, int N_, int lane, char* tile) { x = x_; N = N_; spx = lane >> 3; scol = lane & 7; xs = tile; }
    template <int SLOT> __device__ __forceinline__ void issue(int t, int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = t * 32 + i * 8 + spx;      // (pixels past the end of a ragged last tile read pixel 0: finite values, masked / never stored)
            raw[SLOT][i] = DS_LD(f32x4, x + (size_t)(px < N ? px : 0) * C + c * 32 + scol * 4, DS_BX_SRC0);
        }
    }
    template <int SLOT> __device__ __forceinline__ void stage() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 hi, lo;
            ds_split2(raw[SLOT][i][0], raw[SLOT][i][1], hi.x, lo.x);
            ds_split2(raw[SLOT][i][2], raw[SLOT][i][3], hi.y, lo.y);
            char* const d = xs + (i * 8 + spx) * XS + scol * 8;
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + 64) = lo;
        }
    }
};

// ------------------------------------------------------------------------------------------------ pass 1
template <int NKS, int HB, bool KV, bool Q>
struct P1 {
    static constexpr int C = 16 * NKS, NCH = C / 32, RS = 2 * C + 16;
    static constexpr int NPROJ = (KV ? 2 : 0) + (Q ? 1 : 0), WROWS = NPROJ * HB * 32;          // LDS rows: [k heads][v heads][q heads] of this block
    static constexpr int OFF_WH = 0, OFF_WL = WROWS * RS, OFF_X = 2 * WROWS * RS, OFF_SHQ = OFF_X + NW * XTILE;
    static constexpr int LDS = OFF_SHQ + (Q ? HB * 32 * 4 : 0);
    static_assert(LDS <= 160 * 1024, "pass 1 operands must fit the CU's LDS");
};

template <int NKS, int HB, bool KV, bool Q>
__global__ __launch_bounds__(NT, 1) void attn_x3_pass1_kernel(const ds_attn_x3_params p, const int nseg) {
    using G = P1<NKS, HB, KV, Q>;
    constexpr int C = G::C, NCH = G::NCH, RS = G::RS;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    // XCD-aware block order: hardware block L runs on XCD L % 8.  The 4 / HB head groups of one (segment block, sample) read the same x:
    // they are decoded as consecutive work items of ONE XCD (one L2), so x leaves HBM once.
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, nwg = gx * gy * gz;
    int wid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
    const int hg = wid % gz, sb = (wid / gz) % gx, b = wid / (gz * gx);
    const int h0 = hg * HB;
    // nseg = p.nseg for a launch that writes partials (one per wave); a q-only launch takes its own count (the launcher's: one round of blocks)
    const int ntiles = (p.N + 31) >> 5, per = (ntiles + nseg - 1) / nseg;
    const int seg = sb * NW + wave;
    const int t0 = min(ntiles, seg * per), t1 = min(ntiles, t0 + per);     // an empty segment writes the neutral partial (max = -inf, sum = 0)

    XStream<C> xq;
    char* const xs = sm + G::OFF_X + wave * XTILE;
    xq.init(p.x + (size_t)b * p.N * C, p.N, lane, xs);
    // chunk g + 2 of the wave's chunk sequence (tile t0 chunk 0, 1, .., tile t0 + 1 chunk 0, ..) is requested when chunk g has been staged
    if (t0 < t1) {
        xq.template issue<0>(t0, 0);
        if (NCH > 1) xq.template issue<1>(t0, 1);
    }

    // ---- block prologue: the weight rows of this block's heads, hi plane and lo plane, -> LDS (all requested before the first LDS write)
    {
        const char* const wsrc = reinterpret_cast<const char*>(p.wqkv_hl);
        constexpr int PCS = C / 8;                                       // 16-byte pieces per row
        constexpr int TOT = 2 * G::WROWS * PCS, WIT = TOT / NT;
        static_assert(WIT * NT == TOT, "whole staging iterations");
        u32x4 wst[WIT];
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, plane = i / (G::WROWS * PCS), r = (i / PCS) % G::WROWS, col = i % PCS;
            const int proj = r / (HB * 32), hr = r - proj * (HB * 32);
            const int src = (KV ? (proj == 0 ? 128 : (proj == 1 ? 256 : 0)) : 0) + h0 * 32 + hr;      // rows of [q | k | v] in the packed weights
            wst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wsrc + ((size_t)plane * 384 + src) * C * 2 + col * 16), DS_BX_W);
        }
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, plane = i / (G::WROWS * PCS), r = (i / PCS) % G::WROWS, col = i % PCS;
            *reinterpret_cast<u32x4*>(sm + (plane ? G::OFF_WL : G::OFF_WH) + r * RS + col * 16) = wst[k];
        }
    }
    float ga, gam;
    if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
    else { ga = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB); gam = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB); }
    const float ga2 = ga * LOG2E;
    if constexpr (Q) {
        // additive part of q in the log2 domain, in accumulator order: entry (h, fh, r) = row d = acc_row32(r, fh) of head h0 + h
        float* const shq = reinterpret_cast<float*>(sm + G::OFF_SHQ);
        for (int i = tid; i < HB * 32; i += NT) {
            const int d = (h0 + (i >> 5)) * 32 + acc_row32(i & 15, (i >> 4) & 1);
            shq[i] = LOG2E * (DS_LD(float, p.t1 + d, DS_BX_T1) - gam * DS_LD(float, p.t2 + d, DS_BX_T2) +
                              (p.label_q ? DS_LD(float, p.label_q + (size_t)b * p.lq_stride + d, DS_BX_AUX3) : 0.f));
        }
    }
    float shk2[HB], m[HB], ls[HB];
    f32x16 ctx[HB];
    if constexpr (KV) {
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const int nk = 128 + (h0 + h) * 32 + n;                       // this lane's column d of head h0 + h
            shk2[h] = LOG2E * (DS_LD(float, p.t1 + nk, DS_BX_T1) - gam * DS_LD(float, p.t2 + nk, DS_BX_T2));
            m[h] = -INFINITY;
            ls[h] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctx[h][r] = 0.f;
        }
    }
    __syncthreads();
    const float* const shq = reinterpret_cast<const float*>(sm + G::OFF_SHQ);
    const char* const xf = xs + n * XS + kg * 16;                        // operand fragment (K step ks of the chunk): + ks * 32, lo: + 64
    const char* const wf = sm + G::OFF_WH + n * RS + kg * 16;            // weight fragment: + row0 * RS + (c * 32 + ks * 16) * 2, lo: + OFF_WL
    char* const qp = p.qplanes ? reinterpret_cast<char*>(p.qplanes) + (size_t)b * ntiles * QTILE + lane * 16 : nullptr;

    // one tile; P0 = ring slot of its chunk 0 (compile time: the chunk ring is two register sets)
    auto tile = [&](auto p0c, const int t) {
        constexpr int P0 = decltype(p0c)::value;
        f32x16 acc[G::NPROJ * HB];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // stage chunk c from its slot, then refill the slot with the chunk two ahead in the wave's sequence
            // (the refill is UNCONDITIONAL — past the wave's last tile it re-reads that tile: under `if (t2 < t1)` the loads sit in a conditional
            // region, and the wait for the other slot must then also hold on the path that issued nothing: vmcnt(3..0) instead of vmcnt(4),
            // i.e. every chunk waited for the loads it had just requested)
            const int c2 = c + 2 < NCH ? c + 2 : c + 2 - NCH, t2 = c + 2 < NCH ? t : min(t + 1, t1 - 1);
            // (scheduling fences: left alone, the scheduler sinks the refill loads to the end of the tile — shorter live ranges — and the
            // next stage then waits for loads that have only just been requested)
            __builtin_amdgcn_sched_barrier(0);
            if (((P0 + c) & 1) == 0) {
                xq.template stage<0>();
                xq.template issue<0>(t2, c2);
            } else {
                xq.template stage<1>();
                xq.template issue<1>(t2, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 xh[2], xl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                xh[ks] = *reinterpret_cast<const bf16x8*>(xf + ks * 32);
                xl[ks] = *reinterpret_cast<const bf16x8*>(xf + ks * 32 + 64);
            }
#pragma unroll
            for (int a = 0; a < G::NPROJ * HB; ++a) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wf + a * 32 * RS + (c * 32 + ks * 16) * 2);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wf + G::OFF_WL + a * 32 * RS + (c * 32 + ks * 16) * 2);
                    const bool qproj = Q && a >= (KV ? 2 * HB : 0);
                    if (c == 0 && ks == 0) {
                        const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // literal zero accumulator: no register clearing
                        // k, v: rows = pixels, columns = d (x is the A operand); q: rows = d, columns = pixels (the softmax over d runs over registers)
                        acc[a] = qproj ? mma3(wh, wl, xh[0], xl[0], z) : mma3(xh[0], xl[0], wh, wl, z);
                    } else {
                        acc[a] = qproj ? mma3(wh, wl, xh[ks], xl[ks], acc[a]) : mma3(xh[ks], xl[ks], wh, wl, acc[a]);
                    }
                }
            }
        }
        const int px0 = t * 32;
        const bool ragged = (t == ntiles - 1) && (p.N & 31);             // wave-uniform: only the last tile of a sample
        if constexpr (KV) {
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                f32x16& ak = acc[h];
                const f32x16& av = acc[HB + h];
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (px0 + acc_row32(r, kg) >= p.N) ak[r] = -INFINITY;       // exp2(-inf) = 0
                }
                // online softmax over the pixels in the log2 domain (ga2 > 0: the maximum is taken on the raw accumulators)
                float mr = ak[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mr = fmaxf(mr, ak[r]);
                mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
                const float mn = fmaxf(m[h], fmaf(ga2, mr, shk2[h]));     // finite: a tile holds >= 1 real pixel
                const float sc = exp2_hw(m[h] - mn);                      // m = -inf on the wave's first tile -> 0
                m[h] = mn;
                const float cexp = shk2[h] - mn;
                // v enters the context RAW: its normalisation is affine and is applied to the finished context (see the write-out)
                float P[16], V[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    P[r] = exp2_hw(fmaf(ga2, ak[r], cexp));
                    V[r] = av[r];
                }
                float psum = 0.f, ps1 = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    psum += P[r];
                    ps1 += P[8 + r];
                }
                psum += ps1;
                ls[h] = fmaf(ls[h], sc, psum);
                if (__any(sc != 1.0f)) {                                  // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) ctx[h][r] *= sc;
                }
                bf16x8 ph0, pl0, ph1, pl1, vh0, vl0, vh1, vl1;
                split8(P, ph0, pl0);
                split8(P + 8, ph1, pl1);
                split8(V, vh0, vl0);
                split8(V + 8, vh1, vl1);
                ctx[h] = mma3(vh0, vl0, ph0, pl0, ctx[h]);                // ctx^T[e][d] += V^T P over the tile's pixels (same k order on both sides)
                ctx[h] = mma3(vh1, vl1, ph1, pl1, ctx[h]);
            }
        }
        if constexpr (Q) {
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                const f32x16& aq = acc[(KV ? 2 * HB : 0) + h];
                bf16x8 qh0, ql0, qh1, ql1;
                q_softmax_split(aq, shq + (h * 2 + kg) * 16, ga2, p.scale, qh0, ql0, qh1, ql1);
                char* const dst = qp + (size_t)t * QTILE + ((h0 + h) * 2) * QFRAG;
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(dst), DS_BX_AUX1, qh0);
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(dst + QFRAG), DS_BX_AUX1, qh1);
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(dst + 8 * QFRAG), DS_BX_AUX1, ql0);
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(dst + 9 * QFRAG), DS_BX_AUX1, ql1);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if constexpr (NCH & 1) {
        for (int t = t0; t < t1; t += 2) {
            tile(I0{}, t);
            if (t + 1 < t1) tile(I1{}, t + 1);
        }
    } else {
        for (int t = t0; t < t1; ++t) tile(I0{}, t);
    }
    // ---- this wave's segment of the partials: [32 max (natural log domain)][32 sum][ctx[d][e]] per head
    if constexpr (KV) {
        if (seg >= p.nseg) return;
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            float* out = p.part + (((size_t)b * 4 + h0 + h) * p.nseg + seg) * PARTF;
            const float lsum = ls[h] + __shfl_xor(ls[h], 32, 64);
            if (kg == 0) {
                DS_ST(float, out + n, DS_BX_AUX0, m[h] * (1.0f / LOG2E));
                DS_ST(float, out + 32 + n, DS_BX_AUX0, lsum);
            }
            // v's normalisation, applied once: + shv[e] * sum_px P[px][d].  (All 32 table entries requested before the first store: as one
            // load -> use -> store chain per element the compiler put s_waitcnt vmcnt(0) behind every pair — 32 serial round trips per wave)
            float tv1[16], tv2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nv = 256 + (h0 + h) * 32 + acc_row32(r, kg);
                tv1[r] = DS_LD(float, p.t1 + nv, DS_BX_T1);
                tv2[r] = DS_LD(float, p.t2 + nv, DS_BX_T2);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float shv = tv1[r] - gam * tv2[r];
                DS_ST(float, out + 64 + n * 32 + acc_row32(r, kg), DS_BX_AUX0, fmaf(ga, ctx[h][r], shv * lsum));
            }
        }
    }
}

// form B with out_planes: the block's output also (or only) as hi / lo bf16 planes [B][N][2C] — the input format of the Down / Upsample that
// follows (DS_CONV_F_SPLIT_IN): no ds_split_planes pass over it
__device__ __forceinline__ void x3_store_planes(const ds_attn_x3_params& p, int b, int px, int C, int c, const f32x4& v) {
    bf16* const o2 = reinterpret_cast<bf16*>(p.out_planes) + ((size_t)b * p.N + px) * (2 * C) + c;
    uint2 hi, lo;
    ds_split2(v[0], v[1], hi.x, lo.x);
    ds_split2(v[2], v[3], hi.y, lo.y);
    DS_ST(uint2, reinterpret_cast<uint2*>(o2), DS_BX_AUX0, hi);      // (AUX0 = out_planes in the output launches' bounds table)
    DS_ST(uint2, reinterpret_cast<uint2*>(o2 + C), DS_BX_AUX0, lo);
}

// ------------------------------------------------------------------------------------------------ fold: M_b = Wout . ctx_b^T, split, operand layout
// M_b[c][h*32 + d] = sum_e Wout[c][h*32 + e] ctx[b][h][d][e] in fp32; stored as two bf16 planes [b][plane][C][128] with
//   row  m of channel block cb  <-  channel cb*32 + 16*((m>>2)&1) + (m&3) + 4*(m>>3)      (accumulator register r of lane half fh = channel 16 fh + r)
//   column h*32 + s*16 + kg*8 + j  <-  d = 16 s + 8 (j>>2) + 4 kg + (j&3)                   (the k order of a packed 32x32 accumulator used as B operand)
// Both operands come to LDS in ONE round trip (16-byte rows): with Wout read per output straight from global memory a thread ran 16 dependent
// batches of 32 loads — 13.5 us per launch whatever the batch, 16 launches per forward (5 % of a batch-1 step).
__global__ __launch_bounds__(256) void attn_x3_fold_kernel(const float* ctx, const float* wout, bf16* mfold, int C) {
    constexpr int CP = 36;                                           // row pitch in floats: 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float sctx[4 * 32 * CP];      // [h*32 + d][e]
    __shared__ __attribute__((aligned(16))) float sw[32 * 128];           // Wout rows of this channel block: [channel][h*32 + e]
    const int cb = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    {
        f32x4 cv[4], wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 256;                                 // 1024 pieces of 4 floats each
            cv[k] = *reinterpret_cast<const f32x4*>(ctx + (size_t)b * 4096 + 4 * i);
            wv[k] = *reinterpret_cast<const f32x4*>(wout + (size_t)cb * 32 * 128 + 4 * i);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * 256;
            *reinterpret_cast<f32x4*>(sctx + (i >> 3) * CP + 4 * (i & 7)) = cv[k];
            *reinterpret_cast<f32x4*>(sw + 4 * i) = wv[k];
        }
    }
    __syncthreads();
    for (int o = tid; o < 32 * 128; o += 256) {
        const int mrow = o >> 7, pos = o & 127;
        const int cl = 16 * ((mrow >> 2) & 1) + (mrow & 3) + 4 * (mrow >> 3);
        const int h = pos >> 5, pp = pos & 31, s = pp >> 4, kg = (pp >> 3) & 1, j = pp & 7;
        const int d = 16 * s + 8 * (j >> 2) + 4 * kg + (j & 3);
        const float* w = sw + cl * 128 + h * 32;
        const float* cr = sctx + (h * 32 + d) * CP;
        float acc = 0.f;
#pragma unroll
        for (int e4 = 0; e4 < 8; ++e4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + 4 * e4), c4 = *reinterpret_cast<const f32x4*>(cr + 4 * e4);
            acc = fmaf(w4[0], c4[0], acc); acc = fmaf(w4[1], c4[1], acc); acc = fmaf(w4[2], c4[2], acc); acc = fmaf(w4[3], c4[3], acc);
        }
        const bf16 hi = (bf16)acc;
        const size_t at = ((size_t)b * 2 * C + cb * 32 + mrow) * 128 + pos;
        mfold[at] = hi;
        mfold[at + (size_t)C * 128] = (bf16)(acc - (float)hi);
    }
}

// ------------------------------------------------------------------------------------------------ pass 2
struct Z2 {
    static constexpr int M_RS = 2 * 128 + 16;                         // an odd number of 16-byte slots
    static constexpr int OFF_MH = 0, OFF_ML = 96 * M_RS, OFF_T = 2 * 96 * M_RS, OFF_BIAS = OFF_T + NW * XTILE, OFF_RED = OFF_BIAS + 96 * 4;
    static constexpr int LDS = OFF_RED + 64 + 2 * 96 * 4;            // (+ scale / shift rows of the output GroupNorm, MODE 2)
};

// MODE 0: y and its GroupNorm partials (form A).  MODE 1: the partials alone.  MODE 2: out = x + GroupNorm(y) (statistics from MODE 1's
// partials), y itself never leaves the registers.
template <int MODE>
__global__ __launch_bounds__(NT, 1) void attn_x3_z_kernel(const ds_attn_x3_params p, const int tiles_per_block) {
    using G = Z2;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    float* const sbias = reinterpret_cast<float*>(sm + G::OFF_BIAS);
    float* const red = reinterpret_cast<float*>(sm + G::OFF_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    // the C / 96 channel groups of one (tile range, sample) read the same q planes: consecutive work items of one XCD (see pass 1)
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, nwg = gx * gy * gz;
    int wid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
    const int cg = wid % gz, tb = (wid / gz) % gx, b = wid / (gz * gx);
    const int C = p.C, c0 = cg * 96;
    const int ntiles = (p.N + 31) >> 5;
    const int t0 = tb * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
    const char* const qp = reinterpret_cast<const char*>(p.qplanes) + (size_t)b * ntiles * QTILE + lane * 16;
    float* const yout = (MODE == 2 ? (p.out ? p.out : reinterpret_cast<float*>(p.out_planes)) : p.y) + (size_t)b * p.N * C;   // (MODE 2 without `out`: never stored through)
    const float* const xres = p.x + (size_t)b * p.N * C;

    // B fragments of a tile: fragment f = plane * 8 + head * 2 + s, 1 KB contiguous per instruction
    bf16x8 qf[16];
    auto load_q = [&](int t) {
#pragma unroll
        for (int f = 0; f < 16; ++f) qf[f] = DS_LD(bf16x8, reinterpret_cast<const bf16x8*>(qp + (size_t)t * QTILE + f * QFRAG), DS_BX_AUX1);
    };
    if (t0 + wave < t1) load_q(t0 + wave);
    {
        // this sample's folded to_out matrix, rows c0 .. c0 + 95 of both planes -> LDS
        const char* mb = reinterpret_cast<const char*>(p.mfold) + (size_t)b * 2 * C * 256;
        constexpr int MIT = 2 * 96 * 16 / NT;
        static_assert(MIT * NT == 2 * 96 * 16, "whole staging iterations");
        u32x4 mst[MIT];
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, plane = i / (96 * 16), row = (i >> 4) % 96, col = i & 15;
            mst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(mb + ((size_t)plane * C + c0 + row) * 256 + col * 16), DS_BX_RES);
        }
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, plane = i / (96 * 16), row = (i >> 4) % 96, col = i & 15;
            *reinterpret_cast<u32x4*>(sm + (plane ? G::OFF_ML : G::OFF_MH) + row * G::M_RS + col * 16) = mst[k];
        }
        for (int i = tid; i < 96; i += NT) sbias[i] = DS_LD(float, p.bias_out + c0 + i, DS_BX_BIAS);
        if constexpr (MODE == 2) {
            float oa, oam;
            gn_from_partials(p.stats_part, gx * gz, (double)C * p.N, p.on_eps, b, oa, oam, DS_BX_STATS);
            // out = x + (a y - a mean) gamma_c + beta_c = x + y (a gamma_c) + (beta_c - a mean gamma_c): the bias row becomes the shift,
            // the scale its own row (the accumulators start from the plain bias: y itself is what gets scaled)
            float* const sgam = reinterpret_cast<float*>(sm + G::OFF_RED + 64);
            for (int i = tid; i < 96; i += NT) {
                const float gmm = DS_LD(float, p.on_gamma + c0 + i, DS_BX_AUX2);
                sgam[i] = oa * gmm;
                sgam[96 + i] = DS_LD(float, p.on_beta + c0 + i, DS_BX_SRC1) - oam * gmm;
            }
        }
    }
    __syncthreads();
    const float* const sgam = reinterpret_cast<const float*>(sm + G::OFF_RED + 64);
    const char* const m_l = sm + G::OFF_MH + n * G::M_RS + kg * 16;       // A fragment (block cb, step hs): + cb*32*M_RS + hs*32; lo: + OFF_ML
    char* const tt = sm + G::OFF_T + wave * XTILE;                        // 32 px x 32 ch fp32 transpose tile (144-byte rows)
    const int spx = lane >> 3, scol = lane & 7;
    float s1 = 0.f, s2 = 0.f;
    for (int t = t0 + wave; t < t1; t += NW) {
        bf16x8 qc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) qc[f] = qf[f];
        load_q(t + NW < t1 ? t + NW : t);                                 // next tile's fragments under this tile's products (unconditional: see pass 1)
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            f32x16 Z;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + cb * 32 + 16 * kg + 4 * k);      // accumulators start from the bias
#pragma unroll
                for (int e = 0; e < 4; ++e) Z[4 * k + e] = bv[e];
            }
#pragma unroll
            for (int hs = 0; hs < 8; ++hs) {
                const bf16x8 mh = *reinterpret_cast<const bf16x8*>(m_l + cb * 32 * G::M_RS + hs * 32);
                const bf16x8 ml = *reinterpret_cast<const bf16x8*>(m_l + G::OFF_ML + cb * 32 * G::M_RS + hs * 32);
                Z = mma3(mh, ml, qc[hs], qc[8 + hs], Z);
            }
            if constexpr (MODE == 1) {
                // statistics straight from the accumulators: lane = pixel n of the tile (both halves), 16 channels each
                if (t * 32 + n < p.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s1 += Z[r];
                        s2 = fmaf(Z[r], Z[r], s2);
                    }
                }
                // (fence: with no store between them the three channel blocks are independent, the scheduler hoists all 48 fragment reads
                // to the top of the tile and spills — 644 bytes of scratch per lane, 188 us instead of 90)
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            // lane (pixel n, half kg) holds channels c0 + cb*32 + 16 kg + r: 64 contiguous bytes.  Through the wave's LDS tile the stores
            // become whole 128-byte lines (8 lanes per pixel) — 16-byte pieces from 64 different lines per instruction are bound by the
            // L2 request rate, not by bytes
            f32x4 xr[4];
            if constexpr (MODE == 2) {
                // the residual, exact fp32, on the contiguous side (requested before the transpose)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = t * 32 + i * 8 + spx;
                    xr[i] = DS_LD(f32x4, xres + (size_t)(px < p.N ? px : 0) * C + c0 + cb * 32 + scol * 4, DS_BX_SRC0);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Z[4 * k + e];
                *reinterpret_cast<f32x4*>(tt + n * XS + kg * 64 + k * 16) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pl = i * 8 + spx, px = t * 32 + pl;
                f32x4 v = *reinterpret_cast<const f32x4*>(tt + pl * XS + scol * 16);
                if constexpr (MODE == 2) {
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(sgam + cb * 32 + scol * 4), sh4 = *reinterpret_cast<const f32x4*>(sgam + 96 + cb * 32 + scol * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xr[i][e] + fmaf(v[e], sc4[e], sh4[e]);
                }
                if (px < p.N) {
                    if (MODE != 2 || p.out) DS_ST(f32x4, reinterpret_cast<f32x4*>(yout + (size_t)px * C + c0 + cb * 32 + scol * 4), DS_BX_OUT, v);
                    if constexpr (MODE == 2) {
                        if (p.out_planes) x3_store_planes(p, b, px, C, c0 + cb * 32 + scol * 4, v);
                    }
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s1 += v[e];
                            s2 = fmaf(v[e], v[e], s2);
                        }
                    }
                }
            }
        }
    }
    if constexpr (MODE != 2) {
        if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gx * gz + tb * gz + cg) * 2);
    }
}

// ------------------------------------------------------------------------------------------------ pass 2, fused with the q projection (C = 96)
// Where Wq (hi, lo: 53 KB) and the whole M_b (52 KB) fit LDS together the q planes are not written at all: the block re-reads x (384 B per
// pixel instead of 2 x 512 B of q planes), projects q for the four heads, and multiplies the softmaxed q~ straight into the output channels.
template <int NKS>
struct QZ {
    static constexpr int C = 16 * NKS, NCH = C / 32, CB = C / 32, RS = 2 * C + 16, M_RS = 2 * 128 + 16;
    static constexpr int OFF_WH = 0, OFF_WL = 128 * RS, OFF_MH = 2 * 128 * RS, OFF_ML = OFF_MH + C * M_RS, OFF_X = OFF_ML + C * M_RS;
    static constexpr int OFF_SHQ = OFF_X + NW * XTILE, OFF_BIAS = OFF_SHQ + 128 * 4, OFF_RED = OFF_BIAS + C * 4, OFF_GAM = OFF_RED + 64;
    static constexpr int LDS = OFF_GAM + 2 * C * 4;
    static_assert(LDS <= 160 * 1024, "fused pass 2 operands must fit the CU's LDS");
};

template <int NKS, int MODE>      // MODE as in attn_x3_z_kernel
__global__ __launch_bounds__(NT, 1) void attn_x3_qz_kernel(const ds_attn_x3_params p, const int tiles_per_block) {
    using G = QZ<NKS>;
    constexpr int C = G::C, NCH = G::NCH, RS = G::RS;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    float* const shq = reinterpret_cast<float*>(sm + G::OFF_SHQ);
    float* const sbias = reinterpret_cast<float*>(sm + G::OFF_BIAS);
    float* const red = reinterpret_cast<float*>(sm + G::OFF_RED);
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const int ntiles = (p.N + 31) >> 5;
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
    float* const yout = (MODE == 2 ? (p.out ? p.out : reinterpret_cast<float*>(p.out_planes)) : p.y) + (size_t)b * p.N * C;   // (MODE 2 without `out`: never stored through)

    XStream<C> xq;
    char* const xs = sm + G::OFF_X + wave * XTILE;
    xq.init(p.x + (size_t)b * p.N * C, p.N, lane, xs);
    if (t0 + wave < t1) {
        xq.template issue<0>(t0 + wave, 0);
        xq.template issue<1>(t0 + wave, 1);
    }
    {
        // Wq (rows 0..127 of both planes of the packed qkv weights) and this sample's folded to_out matrix (both planes) -> LDS
        const char* const wsrc = reinterpret_cast<const char*>(p.wqkv_hl);
        const char* const mb = reinterpret_cast<const char*>(p.mfold) + (size_t)b * 2 * C * 256;
        constexpr int PCS = C / 8, WIT = 2 * 128 * PCS / NT, MIT = 2 * C * 16 / NT;
        static_assert(WIT * NT == 2 * 128 * PCS && MIT * NT == 2 * C * 16, "whole staging iterations");
        u32x4 wst[WIT], mst[MIT];
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, plane = i / (128 * PCS), r = (i / PCS) % 128, col = i % PCS;
            wst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wsrc + ((size_t)plane * 384 + r) * C * 2 + col * 16), DS_BX_W);
        }
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, plane = i / (C * 16), row = (i >> 4) % C, col = i & 15;
            mst[k] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(mb + ((size_t)plane * C + row) * 256 + col * 16), DS_BX_RES);
        }
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = tid + k * NT, plane = i / (128 * PCS), r = (i / PCS) % 128, col = i % PCS;
            *reinterpret_cast<u32x4*>(sm + (plane ? G::OFF_WL : G::OFF_WH) + r * RS + col * 16) = wst[k];
        }
#pragma unroll
        for (int k = 0; k < MIT; ++k) {
            const int i = tid + k * NT, plane = i / (C * 16), row = (i >> 4) % C, col = i & 15;
            *reinterpret_cast<u32x4*>(sm + (plane ? G::OFF_ML : G::OFF_MH) + row * G::M_RS + col * 16) = mst[k];
        }
        float ga, gam;
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
        else { ga = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB); gam = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB); }
        for (int i = tid; i < 128; i += NT) {
            const int d = (i >> 5) * 32 + acc_row32(i & 15, (i >> 4) & 1);
            shq[i] = LOG2E * (DS_LD(float, p.t1 + d, DS_BX_T1) - gam * DS_LD(float, p.t2 + d, DS_BX_T2) +
                              (p.label_q ? DS_LD(float, p.label_q + (size_t)b * p.lq_stride + d, DS_BX_AUX3) : 0.f));
        }
        for (int i = tid; i < C; i += NT) sbias[i] = DS_LD(float, p.bias_out + i, DS_BX_BIAS);
        if (tid == 0) red[15] = ga * LOG2E;
        if constexpr (MODE == 2) {
            float oa, oam;
            gn_from_partials(p.stats_part, gridDim.x, (double)C * p.N, p.on_eps, b, oa, oam, DS_BX_STATS);
            float* const sg = reinterpret_cast<float*>(sm + G::OFF_GAM);      // out = x + y (a gamma_c) + (beta_c - a mean gamma_c)
            for (int i = tid; i < C; i += NT) {
                const float gmm = DS_LD(float, p.on_gamma + i, DS_BX_AUX2);
                sg[i] = oa * gmm;
                sg[C + i] = DS_LD(float, p.on_beta + i, DS_BX_SRC1) - oam * gmm;
            }
        }
    }
    __syncthreads();
    const float ga2 = red[15];
    __syncthreads();                       // (red is reused by the statistics reduction at the end)
    const char* const xf = xs + n * XS + kg * 16;
    const char* const wf = sm + G::OFF_WH + n * RS + kg * 16;
    const char* const m_l = sm + G::OFF_MH + n * G::M_RS + kg * 16;
    const int spx = lane >> 3, scol = lane & 7;
    const float* const sgam = reinterpret_cast<const float*>(sm + G::OFF_GAM);
    float s1 = 0.f, s2 = 0.f;

    auto tile = [&](auto p0c, const int t) {
        constexpr int P0 = decltype(p0c)::value;
        f32x16 aq[4];
        // MODE 2: the tile's raw fp32 pieces are KEPT for the residual — they arrive in exactly the epilogue's store pattern (pixel i * 8 + spx,
        // channels 32 c + 4 scol); re-reading them there cost a second trip through the fabric (PMC: 1.6 GB read per launch for a 0.8 GB x)
        f32x4 xkeep[MODE == 2 ? NCH : 1][4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int c2 = c + 2 < NCH ? c + 2 : c + 2 - NCH, t2 = c + 2 < NCH ? t : (t + NW < t1 ? t + NW : t);      // unconditional refill (see pass 1)
            // (scheduling fences: left alone, the scheduler sinks the refill loads to the end of the tile — shorter live ranges — and the
            // next stage then waits for loads that have only just been requested)
            __builtin_amdgcn_sched_barrier(0);
            if (((P0 + c) & 1) == 0) {
                xq.template stage<0>();
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xkeep[c][i] = xq.raw[0][i];
                }
                xq.template issue<0>(t2, c2);
            } else {
                xq.template stage<1>();
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xkeep[c][i] = xq.raw[1][i];
                }
                xq.template issue<1>(t2, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 xh[2], xl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                xh[ks] = *reinterpret_cast<const bf16x8*>(xf + ks * 32);
                xl[ks] = *reinterpret_cast<const bf16x8*>(xf + ks * 32 + 64);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wf + h * 32 * RS + (c * 32 + ks * 16) * 2);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wf + G::OFF_WL + h * 32 * RS + (c * 32 + ks * 16) * 2);
                    if (c == 0 && ks == 0) {
                        const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        aq[h] = mma3(wh, wl, xh[0], xl[0], z);
                    } else {
                        aq[h] = mma3(wh, wl, xh[ks], xl[ks], aq[h]);
                    }
                }
            }
        }
        bf16x8 qh[8], ql[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) q_softmax_split(aq[h], shq + (h * 2 + kg) * 16, ga2, p.scale, qh[2 * h], ql[2 * h], qh[2 * h + 1], ql[2 * h + 1]);
        // ---- Z[c][px] = sum_{h,d} M_b[c][h*32 + d] q~_h[d][px] + bias[c]; the wave's x tile is consumed: it is the transpose tile now
#pragma unroll
        for (int cb = 0; cb < G::CB; ++cb) {
            f32x16 Z;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + cb * 32 + 16 * kg + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) Z[4 * k + e] = bv[e];
            }
#pragma unroll
            for (int hs = 0; hs < 8; ++hs) {
                const bf16x8 mh = *reinterpret_cast<const bf16x8*>(m_l + cb * 32 * G::M_RS + hs * 32);
                const bf16x8 ml = *reinterpret_cast<const bf16x8*>(m_l + (G::OFF_ML - G::OFF_MH) + cb * 32 * G::M_RS + hs * 32);
                Z = mma3(mh, ml, qh[hs], ql[hs], Z);
            }
            if constexpr (MODE == 1) {
                if (t * 32 + n < p.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s1 += Z[r];
                        s2 = fmaf(Z[r], Z[r], s2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            f32x4 xr[4];
            if constexpr (MODE == 2) {
                // the residual, exact fp32 (the staged tile holds hi + lo = x to 2^-17 only): the pieces kept above
                static_assert(G::CB == NCH, "one 32-channel output block per input chunk");
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[i] = xkeep[cb][i];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Z[4 * k + e];
                *reinterpret_cast<f32x4*>(xs + n * XS + kg * 64 + k * 16) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pl = i * 8 + spx, px = t * 32 + pl;
                f32x4 v = *reinterpret_cast<const f32x4*>(xs + pl * XS + scol * 16);
                if constexpr (MODE == 2) {
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(sgam + cb * 32 + scol * 4), sh4 = *reinterpret_cast<const f32x4*>(sgam + C + cb * 32 + scol * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xr[i][e] + fmaf(v[e], sc4[e], sh4[e]);
                }
                if (px < p.N) {
                    if (MODE != 2 || p.out) DS_ST(f32x4, reinterpret_cast<f32x4*>(yout + (size_t)px * C + cb * 32 + scol * 4), DS_BX_OUT, v);
                    if constexpr (MODE == 2) {
                        if (p.out_planes) x3_store_planes(p, b, px, C, cb * 32 + scol * 4, v);
                    }
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s1 += v[e];
                            s2 = fmaf(v[e], v[e], s2);
                        }
                    }
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if constexpr (NCH & 1) {
        for (int t = t0 + wave; t < t1; t += 2 * NW) {
            tile(I0{}, t);
            if (t + NW < t1) tile(I1{}, t + NW);
        }
    } else {
        for (int t = t0 + wave; t < t1; t += NW) tile(I0{}, t);
    }
    if constexpr (MODE != 2) {
        if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
    }
}

__global__ void pack_attn_x3_kernel(const float* wqkv, const float* gamma, bf16* out, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 384 * C) {
        const float w = wqkv[i] * gamma[i % C];
        const bf16 hi = (bf16)w;
        out[i] = hi;
        out[(size_t)384 * C + i] = (bf16)(w - (float)hi);
    }
}

int check(const ds_attn_x3_params* p) {
    DS_REQUIRE(p && p->x && p->wqkv_hl && p->t1 && p->t2 && (p->gn_ab || p->gn_part) && p->part && p->ctx && p->mfold && (p->qplanes || p->C == 96),
               "attn_x3: null pointer");
    DS_REQUIRE(p->C == 96 || p->C == 192 || p->C == 384, "attn_x3: C=%d unsupported (96, 192, 384)", p->C);
    DS_REQUIRE(p->B > 0 && p->N > 0 && p->nseg > 0, "attn_x3: bad sizes");
    if (!ds_aligned16(p->x) || !ds_aligned16(p->wqkv_hl) || (p->qplanes && !ds_aligned16(p->qplanes)) || !ds_aligned16(p->mfold))
        DS_FAIL(DS_EALIGN, "attn_x3: pointers must be 16-byte aligned");
    return DS_OK;
}

int z_tiles_per_block(int N, int B, int C) {
    // blocks per (sample, channel group): every CU busy with few, long-lived blocks (a block stages 52 KB of M_b), at least one tile per wave
    // (C = 96: the fused q + Z kernel, one group)
    const int ntiles = (N + 31) / 32, groups = C / 96;
    int nb = (256 + B * groups - 1) / (B * groups);
    const int max_nb = (ntiles + NW - 1) / NW;
    if (nb > max_nb) nb = max_nb;
    if (nb < 1) nb = 1;
    return (ntiles + nb - 1) / nb;
}

#if DS_BOUNDS
void x3_publish_bounds(const ds_attn_x3_params* p, int kernel, int stats_parts, hipStream_t st) {
    const long long ntiles = (p->N + 31) / 32;
    DsBxHost h(kernel);
    h.set(DS_BX_SRC0, p->x, (long long)p->B * p->N * p->C * 4);
    h.set(DS_BX_W, p->wqkv_hl, (long long)2 * 384 * p->C * 2);
    h.set(DS_BX_T1, p->t1, 384 * 4).set(DS_BX_T2, p->t2, 384 * 4);
    h.set(DS_BX_GNAB, p->gn_ab, (long long)p->B * 2 * 4);
    h.set(DS_BX_GNPART, p->gn_part, (long long)p->B * p->gn_parts * 2 * 4);
    if (kernel == DS_K_ATTN_OUT && p->out_planes) h.set(DS_BX_AUX0, p->out_planes, (long long)p->B * p->N * p->C * 4);
    else h.set(DS_BX_AUX0, p->part, (long long)p->B * 4 * p->nseg * PARTF * 4);
    h.set(DS_BX_AUX1, p->qplanes, (long long)p->B * ntiles * QTILE);
    h.set(DS_BX_AUX3, p->label_q, p->label_q ? ((long long)(p->B - 1) * p->lq_stride + 128) * 4 : 0);
    h.set(DS_BX_BIAS, p->bias_out, (long long)p->C * 4);
    h.set(DS_BX_OUT, p->out ? p->out : p->y, (long long)p->B * p->N * p->C * 4);
    h.set(DS_BX_AUX2, p->on_gamma, (long long)p->C * 4).set(DS_BX_SRC1, p->on_beta, (long long)p->C * 4);
    h.set(DS_BX_STATS, p->stats_part, (long long)p->B * stats_parts * 2 * 4);
    h.set(DS_BX_RES, p->mfold, (long long)p->B * 2 * p->C * 256);
    h.publish(st);
}
#endif

template <int NKS, int HB, bool KV, bool Q>
int launch_pass1(const ds_attn_x3_params* p, hipStream_t st) {
    using G = P1<NKS, HB, KV, Q>;
    auto kern = attn_x3_pass1_kernel<NKS, HB, KV, Q>;
    DS_SET_MAX_LDS(kern, G::LDS, "attn_x3_pass1");
    // a k / v launch needs a wave for every segment of partials; a q-only launch cuts the tiles into its own one round of blocks
    int nseg = p->nseg;
    if (!KV) {
        const int ntiles = (p->N + 31) / 32;
        nseg = (2048 / (4 / HB) + p->B - 1) / p->B;
        nseg = (nseg + NW - 1) / NW * NW;
        if (nseg > ntiles) nseg = ntiles;
    }
    hipLaunchKernelGGL(kern, dim3((nseg + NW - 1) / NW, p->B, 4 / HB), dim3(NT), G::LDS, st, *p, nseg);
    DS_CHECK_LAUNCH("attn_x3_pass1");
    return DS_OK;
}

}  // namespace

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_attn_x3(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" int ds_pack_attn_x3(const float* wqkv, const float* gamma, void* wqkv_hl, int C, void* stream) {
    DS_REQUIRE(wqkv && gamma && wqkv_hl && C > 0, "pack_attn_x3: bad args");
    hipLaunchKernelGGL(pack_attn_x3_kernel, dim3((384 * C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wqkv, gamma,
                       reinterpret_cast<bf16*>(wqkv_hl), C);
    DS_CHECK_LAUNCH("pack_attn_x3");
    return DS_OK;
}

extern "C" int ds_attn_x3_segments(int B, int N, int C) {
    // one segment per wave, one round of blocks: 256 CUs x 8 waves over B samples x (4 / HB) head groups of the k / v launch — but never
    // fewer segments than a block has waves (a block stages 50 - 100 KB of weights: idle waves are the dearer waste)
    const int ntiles = (N + 31) / 32, groups = C == 384 ? 4 : 2;
    int s = (2048 / groups) / (B > 0 ? B : 1);
    if (s < NW) s = NW;
    if (s > 128) s = 128;                                       // (small batches: up to 16 blocks per sample and head group; the combine walks the segments serially, in eights: 256 cost it 21 us at batch 1)
    if (s > ntiles) s = ntiles;
    return s < 1 ? 1 : s;
}

extern "C" size_t ds_attn_x3_qplane_bytes(int B, int N) { return (size_t)B * ((N + 31) / 32) * QTILE; }
extern "C" size_t ds_attn_x3_mfold_bytes(int B, int C) { return (size_t)B * 2 * C * 256; }

extern "C" int ds_attn_x3_stats_parts(const ds_attn_x3_params* p) {
    const int ntiles = (p->N + 31) / 32, per = z_tiles_per_block(p->N, p->B, p->C);
    return ((ntiles + per - 1) / per) * (p->C / 96);
}

extern "C" int ds_attn_x3_context(const ds_attn_x3_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    x3_publish_bounds(p, DS_K_ATTN_CTX, 0, st);
#endif
    if (p->C == 96) rc = launch_pass1<6, 2, true, false>(p, st);        // (q is projected again by the fused pass 2: no q planes at C = 96)
    else if (p->C == 192) {
        // two launches: k / v for two heads per block (x split twice) and q for all four (once) — as one k / v / q launch the weights of
        // ONE head fill LDS (77 KB), x is staged and split by four head groups, and the kernel is bound by its vector-instruction count
        static const bool one = getenv("DS_X3_ATTN_192_ONE") != nullptr;      // A/B switch
        if (one) rc = launch_pass1<12, 1, true, true>(p, st);
        else {
            rc = launch_pass1<12, 2, true, false>(p, st);
            if (!rc) rc = launch_pass1<12, 4, false, true>(p, st);
        }
    } else {
        rc = launch_pass1<24, 1, true, false>(p, st);
        if (!rc) rc = launch_pass1<24, 2, false, true>(p, st);
    }
    if (rc) return rc;
    ds_attn_params q;
    memset(&q, 0, sizeof(q));
    q.B = p->B; q.N = p->N; q.heads = 4; q.nseg = p->nseg; q.part = p->part; q.ctx = p->ctx;
    return ds_linattn_launch_combine(&q, st);
}

extern "C" int ds_attn_x3_output(const ds_attn_x3_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    const bool formb = p->out != nullptr || p->out_planes != nullptr;
    DS_REQUIRE(p->wout && p->bias_out && (formb || p->y), "attn_x3_output: null pointer");
    DS_REQUIRE(!formb || (p->stats_part && p->on_gamma && p->on_beta && ds_aligned16(p->out) && ds_aligned16(p->out_planes)),
               "attn_x3_output: form B needs out and / or out_planes (16-byte aligned), stats_part, on_gamma, on_beta");
    DS_REQUIRE(formb || ds_aligned16(p->y), "attn_x3_output: y must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ntiles = (p->N + 31) / 32, per = z_tiles_per_block(p->N, p->B, p->C), nb = (ntiles + per - 1) / per;
#if DS_BOUNDS
    x3_publish_bounds(p, DS_K_ATTN_OUT, nb * (p->C / 96), st);
#endif
    hipLaunchKernelGGL(attn_x3_fold_kernel, dim3(p->C / 32, p->B), dim3(256), 0, st, p->ctx, p->wout, reinterpret_cast<bf16*>(p->mfold), p->C);
    DS_CHECK_LAUNCH("attn_x3_fold");
#define DS_X3_QZ(MODE_)                                                                                         \
    do {                                                                                                        \
        DS_SET_MAX_LDS((attn_x3_qz_kernel<6, MODE_>), QZ<6>::LDS, "attn_x3_qz");                                \
        hipLaunchKernelGGL((attn_x3_qz_kernel<6, MODE_>), dim3(nb, p->B), dim3(NT), QZ<6>::LDS, st, *p, per);   \
        DS_CHECK_LAUNCH("attn_x3_qz");                                                                          \
    } while (0)
#define DS_X3_Z(MODE_)                                                                                              \
    do {                                                                                                            \
        DS_SET_MAX_LDS(attn_x3_z_kernel<MODE_>, Z2::LDS, "attn_x3_z");                                              \
        hipLaunchKernelGGL(attn_x3_z_kernel<MODE_>, dim3(nb, p->B, p->C / 96), dim3(NT), Z2::LDS, st, *p, per);     \
        DS_CHECK_LAUNCH("attn_x3_z");                                                                               \
    } while (0)
    if (p->C == 96) {
        if (formb) { DS_X3_QZ(1); DS_X3_QZ(2); }
        else DS_X3_QZ(0);
    } else {
        if (formb) { DS_X3_Z(1); DS_X3_Z(2); }
        else DS_X3_Z(0);
    }
#undef DS_X3_QZ
#undef DS_X3_Z
    return DS_OK;
}
