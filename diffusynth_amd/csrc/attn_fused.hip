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

// ------------------------------------------------------------------------------------------------ pass 1
template <int NKS, bool WREG>
__global__ __launch_bounds__(256) void attn_fused_ctx_kernel(const ds_attn_fused_params p) {
    constexpr int C = NKS * 16;
    const int seg = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, head = threadIdx.x >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int ntiles = (p.N + 31) / 32, per = (ntiles + p.nseg - 1) / p.nseg;
    const int t0 = seg * per, t1 = min(ntiles, t0 + per);
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    const int nk = 128 + head * 32 + frow, nv = 256 + head * 32 + frow;
    const bf16* wk = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)nk * C + fh * 8;
    const bf16* wv = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)nv * C + fh * 8;
    float ga, gam;
    if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
    else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
    const float shk = p.t1[nk] - gam * p.t2[nk], shv = p.t1[nv] - gam * p.t2[nv];

    bf16x8 Wk[WREG ? NKS : 1], Wv[WREG ? NKS : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Wk[ks] = *reinterpret_cast<const bf16x8*>(wk + ks * 16);
            Wv[ks] = *reinterpret_cast<const bf16x8*>(wv + ks * 16);
        }
    }
    // sweep 1: per-d maximum of k over the segment
    float mx = -INFINITY;
    for (int t = t0; t < t1; ++t) {
        const int px = min(t * 32 + frow, p.N - 1);
        const bf16* xr = x + (size_t)px * C + fh * 8;
        f32x16 ak;
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
            const bf16x8 w = WREG ? Wk[WREG ? ks : 0] : *reinterpret_cast<const bf16x8*>(wk + ks * 16);
            ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, ak, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (t * 32 + acc_row(r, fh) < p.N) mx = fmaxf(mx, ga * ak[r] + shk);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // sweep 2: P = exp(k - max), l += sum P, ctx += P^T V
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    float ls = 0.f;
    for (int t = t0; t < t1; ++t) {
        const int px = min(t * 32 + frow, p.N - 1);
        const bf16* xr = x + (size_t)px * C + fh * 8;
        f32x16 ak, av;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ak[r] = 0.f;
            av[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
            const bf16x8 w1 = WREG ? Wk[WREG ? ks : 0] : *reinterpret_cast<const bf16x8*>(wk + ks * 16);
            const bf16x8 w2 = WREG ? Wv[WREG ? ks : 0] : *reinterpret_cast<const bf16x8*>(wv + ks * 16);
            ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w1, ak, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w2, av, 0, 0, 0);
        }
        float P[16], V[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = t * 32 + acc_row(r, fh) < p.N;
            P[r] = ok ? __expf(ga * ak[r] + shk - mx) : 0.f;
            V[r] = ga * av[r] + shv;
            ls += P[r];
        }
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(P), pack8(V), ctx, 0, 0, 0);
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(P + 8), pack8(V + 8), ctx, 0, 0, 0);
    }
    ls += __shfl_xor(ls, 32, 64);
    float* out = p.part + (((size_t)b * 4 + head) * p.nseg + seg) * PARTF;
    if (fh == 0) {
        out[frow] = mx;          // lane = d
        out[32 + frow] = ls;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[64 + acc_row(r, fh) * 32 + frow] = ctx[r];   // ctx[d][e], e on the lane
}

// ------------------------------------------------------------------------------------------------ pass 2
// 32-pixel tiles per wave: per-sample constants (ctx^T fragments, q shifts) are built once per block; large images
// amortise them over 4 tiles, small ones keep 1 so that the grid still fills the chip (function of N only)
static inline int out_tpw(int N) { return N >= 4096 ? 4 : 1; }

template <int NKS>
__global__ __launch_bounds__(256, NKS >= 24 ? 1 : 2) void attn_fused_out_kernel(const ds_attn_fused_params p, int tpw) {
    constexpr int C = NKS * 16, CB = C / 32;
    constexpr int CG = CB < 3 ? CB : 3;          // c-blocks staged per store group (<= 96 channels)
    constexpr int SW = CG * 32 + 4;
    extern __shared__ __attribute__((aligned(16))) float smf[];   // stage[4][32][SW] | shq[128] | ctxA[4][2][64] x 16 B
    __shared__ __attribute__((aligned(16))) float red[8];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    float* stage = smf + wave * (32 * SW);
    float* shq = smf + 4 * 32 * SW;
    bf16x8* ctxA = reinterpret_cast<bf16x8*>(shq + 128);
    const int ntiles = (p.N + 31) / 32;
    const bf16* x = reinterpret_cast<const bf16*>(p.x) + (size_t)b * p.N * C;
    bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)b * p.N * C;
    float ga, gam;
    if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
    else { ga = p.gn_ab[2 * b]; gam = p.gn_ab[2 * b + 1]; }
    // ---- per-sample constants -> LDS: additive part of q (fold shift + label_q), ctx^T as permuted-k A fragments
    if (threadIdx.x < 128) {
        const int d = threadIdx.x;
        float v = p.t1[d] - gam * p.t2[d];
        if (p.label_q) v += p.label_q[(size_t)b * p.lq_stride + d];
        shq[d] = v;
    }
    {
        const float* ctx = p.ctx + ((size_t)b * 4 + wave) * 1024;     // this wave prepares head `wave`
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float ca[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ca[j] = ctx[(16 * s + 8 * (j >> 2) + 4 * fh + (j & 3)) * 32 + frow];
            ctxA[(wave * 2 + s) * 64 + lane] = pack8(ca);
        }
    }
    __syncthreads();

    float s1 = 0.f, s2 = 0.f;
    const int tile0 = (blockIdx.x * 4 + wave) * tpw;
    for (int ti = 0; ti < tpw; ++ti) {
        const int tile = tile0 + ti;
        if (tile >= ntiles) break;                       // wave-uniform
        const int px = min(tile * 32 + frow, p.N - 1);
        const bf16* xr = x + (size_t)px * C + fh * 8;
        bf16x8 xB[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xB[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
        bf16x8 yB[4][2];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            // q^T tile: rows d (registers), columns = pixels (lanes)
            const bf16* wq = reinterpret_cast<const bf16*>(p.wqkv) + (size_t)(hh * 32 + frow) * C + fh * 8;
            f32x16 aq;
#pragma unroll
            for (int r = 0; r < 16; ++r) aq[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(wq + ks * 16), xB[ks], aq, 0, 0, 0);
            float q[16];
            float mxq = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // rows 8g + 4fh + 0..3
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shq + hh * 32 + 8 * g + 4 * fh);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[4 * g + i] = ga * aq[4 * g + i] + sh[i];
                    mxq = fmaxf(mxq, q[4 * g + i]);
                }
            }
            mxq = fmaxf(mxq, __shfl_xor(mxq, 32, 64));
            float sq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                q[r] = __expf(q[r] - mxq);
                sq += q[r];
            }
            sq += __shfl_xor(sq, 32, 64);
            const float inv = p.scale / sq;
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] *= inv;
            // Y_h[e][px] = sum_d ctx[d][e] q~[d][px]: A = ctx^T in the permuted k order of the accumulator operand
            f32x16 Y;
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[r] = 0.f;
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ctxA[(hh * 2 + 0) * 64 + lane], pack8(q), Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ctxA[(hh * 2 + 1) * 64 + lane], pack8(q + 8), Y, 0, 0, 0);
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = Y[r];
            yB[hh][0] = pack8(yv);
            yB[hh][1] = pack8(yv + 8);
        }
        // Z[c][px] = sum_{h,e} Wout[c][h*32+e] Y_h[e][px] + bias[c]; stored through the LDS stage in groups of <= 96 channels
#pragma unroll
        for (int g0 = 0; g0 < CB; g0 += CG) {
#pragma unroll
            for (int cb = 0; cb < CG; ++cb) {
                if (g0 + cb < CB) {
                    const bf16* wo = reinterpret_cast<const bf16*>(p.wout_perm) + (size_t)((g0 + cb) * 32 + frow) * 128 + fh * 8;
                    f32x16 Z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) Z[r] = 0.f;
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(wo + hh * 32 + s * 16), yB[hh][s], Z, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = (g0 + cb) * 32 + 8 * g + 4 * fh;
                        const f32x4 bo = *reinterpret_cast<const f32x4*>(p.bias_out + c0);
                        f32x4 o = {Z[4 * g] + bo[0], Z[4 * g + 1] + bo[1], Z[4 * g + 2] + bo[2], Z[4 * g + 3] + bo[3]};
                        *reinterpret_cast<f32x4*>(stage + frow * SW + cb * 32 + 8 * g + 4 * fh) = o;   // row = this lane's pixel
                    }
                }
            }
            constexpr int dummy_ = 0;
            (void)dummy_;
            const int gw = (CB - g0 < CG ? CB - g0 : CG) * 32;      // channels in this group
            const int cpr = gw / 8;                                   // 16-byte chunks per pixel row
            for (int slot = lane; slot < 32 * cpr; slot += 64) {
                const int row = slot / cpr, cv = slot - row * cpr;
                const int pxo = tile * 32 + row;
                float v[8];
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(stage + row * SW + cv * 8);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(stage + row * SW + cv * 8 + 4);
                v[0] = u0[0]; v[1] = u0[1]; v[2] = u0[2]; v[3] = u0[3]; v[4] = u1[0]; v[5] = u1[1]; v[6] = u1[2]; v[7] = u1[3];
                if (pxo < p.N) {
                    Vec16<bf16>::store(yout + (size_t)pxo * C + g0 * 32 + cv * 8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s1 += v[j];
                        s2 += v[j] * v[j];
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

extern "C" int ds_attn_fused_context(const ds_attn_fused_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(p->nseg, p->B);
    if (p->C == 96) hipLaunchKernelGGL((attn_fused_ctx_kernel<6, true>), grid, dim3(256), 0, st, *p);
    else if (p->C == 192) hipLaunchKernelGGL((attn_fused_ctx_kernel<12, true>), grid, dim3(256), 0, st, *p);
    else hipLaunchKernelGGL((attn_fused_ctx_kernel<24, false>), grid, dim3(256), 0, st, *p);
    DS_CHECK_LAUNCH("attn_fused_ctx");
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
    const int ntiles = (p->N + 31) / 32;
    const int tpw = out_tpw(p->N);
    dim3 grid((ntiles + 4 * tpw - 1) / (4 * tpw), p->B);
    const int CB = p->C / 32, CG = CB < 3 ? CB : 3;
    const size_t lds = (size_t)4 * 32 * (CG * 32 + 4) * sizeof(float) + 128 * sizeof(float) + 4 * 2 * 64 * 16;
    if (p->C == 96) hipLaunchKernelGGL(attn_fused_out_kernel<6>, grid, dim3(256), lds, st, *p, tpw);
    else if (p->C == 192) hipLaunchKernelGGL(attn_fused_out_kernel<12>, grid, dim3(256), lds, st, *p, tpw);
    else hipLaunchKernelGGL(attn_fused_out_kernel<24>, grid, dim3(256), lds, st, *p, tpw);
    DS_CHECK_LAUNCH("attn_fused_out");
    return DS_OK;
}

extern "C" int ds_attn_fused_stats_parts(const ds_attn_fused_params* p) { const int tpw = out_tpw(p->N); return ((p->N + 31) / 32 + 4 * tpw - 1) / (4 * tpw); }
