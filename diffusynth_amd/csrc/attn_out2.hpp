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
        const int px = t * 32 + n;
        const bool ok = px < p.N;
        const bf16* row = x + (size_t)(ok ? px : 0) * C + kg * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 v = DS_LD(bf16x8, row + ks * 16, DS_BX_SRC0);
            xf[ks] = ok ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    if (t0 + wave < t1) load_x(t0 + wave);

    // ---- block prologue: Wq rows 0..127 of the packed qkv weights and this sample's folded to_out matrix -> LDS
    {
        const char* wq = reinterpret_cast<const char*>(p.wqkv);
        for (int i = tid; i < 128 * 2 * NKS; i += NT) {
            const int row = i / (2 * NKS), col = i - row * (2 * NKS);
            *reinterpret_cast<u32x4*>(sm + G::OFF_WQ + row * G::WQ_RS + col * 16) = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wq + ((size_t)row * C * 2 + col * 16)), DS_BX_W);
        }
        const char* mb = reinterpret_cast<const char*>(mfold) + (size_t)b * C * 256;
        for (int i = tid; i < C * 16; i += NT) {
            const int row = i >> 4, col = i & 15;
            *reinterpret_cast<u32x4*>(sm + G::OFF_M + row * G::M_RS + col * 16) = DS_LD(u32x4, reinterpret_cast<const u32x4*>(mb + ((size_t)row * 256 + col * 16)), DS_BX_RES);
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
#pragma unroll
            for (int r = 0; r < 16; ++r) aq[r] = 0.f;
            // (scheduling fences between the heads: left alone, the scheduler hoists every fragment read of the tile to its top and spills)
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 wf[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wq_l + h * 32 * G::WQ_RS + ks * 32);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], aq, 0, 0, 0);
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
            f32x16 Z;
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 mf[8];
#pragma unroll
                for (int hs = 0; hs < 8; ++hs) mf[hs] = *reinterpret_cast<const bf16x8*>(m_l + cb * 32 * G::M_RS + hs * 32);
#pragma unroll
                for (int hs = 0; hs < 8; ++hs) Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf[hs], qB[hs >> 1][hs & 1], Z, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + cb * 32 + 16 * kg + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * k + e] = Z[4 * k + e] + bv[e];
            }
            if (okp) {
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(yrow + cb * 32), DS_BX_OUT, pack8f(v));
                DS_ST(bf16x8, reinterpret_cast<bf16x8*>(yrow + cb * 32 + 8), DS_BX_OUT, pack8f(v + 8));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s1 += v[r];
                    s2 = fmaf(v[r], v[r], s2);
                }
            }
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

}  // namespace

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
