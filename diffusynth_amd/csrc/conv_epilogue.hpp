// Shared epilogue of the MFMA convolution kernels (gfx950).
//
// 32x32 MFMA accumulators hold, per lane, ONE output channel (lane & 31) and 16 rows (pixels).
// Storing them directly means one 2-byte (bf16) element per lane per instruction — sub-dword
// stores that the memory pipeline handles at ~2 B/clk/CU.  Instead each wave transposes its
// 32-row slab through a private LDS stage (fp32, after GroupNorm fold / bias / activation) and
// then streams it out row-major: every lane owns 16 contiguous output bytes, a wave instruction
// writes whole channel rows, the residual is fetched with the same 16-byte accesses, and the
// (sum, sumsq) partials are taken from the exact stored values.
#pragma once
#include "common.hpp"
#ifndef DS_ABLATE
#define DS_ABLATE 0
#endif
#ifndef DS_EPI_ABL
#define DS_EPI_ABL 0   // diagnostic builds of the register epilogue: bit0 no output stores, bit1 no residual loads, bit2 no activation
#endif

struct ConvCoord {
    bool ok;      // row is a real output pixel
    int ho, wo;   // output coordinates (border class of the GroupNorm fold)
    int pix;      // pixel index inside the sample's output image
};

template <typename T> __device__ __forceinline__ void store_scalar(T* p, float v);
template <> __device__ __forceinline__ void store_scalar<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_scalar<bf16>(bf16* p, float v) { *p = (bf16)v; }

template <int ACT> __device__ __forceinline__ float act_const(float v) {
    if constexpr (ACT == DS_ACT_GELU) return gelu_fast(v);
    else return v;
}

// stage: wave-private LDS, 32 * (FN*32 + 4) floats.  n_base = first output channel of this wave's
// slab, ml_base = first tile-local row of this wave's slab.
// ACT / NCLS9 are compile-time so the per-element code is a handful of instructions: with run-time
// switches inside the 96-element loops the epilogue of a K=96 1x1 convolution took 9x longer than its K loop.
// Output channels are stored in 16-byte groups: Cout is rounded up to the vector width (the extra channels
// are exact zeros because their packed weight rows are zero) and out_C / out_c0 must be vector multiples.
template <typename T, int FM, int FN, int ACT, bool NCLS9, typename CoordFn>
__device__ __forceinline__ void conv_epilogue_body(const ds_conv_params& p, f32x16 (&acc)[FM][FN], int b, int n_base, int ml_base,
                                                   int outHW, float* stage, CoordFn coord, float& s1, float& s2, float gn_a = 1.f,
                                                   float gn_am = 0.f) {
    constexpr int V = Vec16<T>::N;
    constexpr int TN = FN * 32;
    constexpr int SW = TN + 4;  // stage row stride in floats (keeps 16-B alignment, spreads banks)
    const int lane = threadIdx.x & 63, frow = lane & 31, fh = lane >> 5;
    const bool fold = p.gn_ab != nullptr || p.gn_part != nullptr;
    float ga = 1.f, gam = 0.f;
    if (p.gn_part) {
        ga = gn_a;      // reduced from the producer's partials at kernel start (conv_gn_prologue)
        gam = gn_am;
    } else if (p.gn_ab) {
        ga = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
        gam = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
    }
    constexpr int cls_mid = NCLS9 ? 4 : 0;
    float shift_mid[FN];
    bool nok[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n_base + j * 32 + frow;
        nok[j] = n < p.Cout;
        float sv = 0.f;
        if (nok[j]) {
            if (fold) sv = DS_LD(float, p.fold_t1 + cls_mid * p.Cout + n, DS_BX_T1) - gam * DS_LD(float, p.fold_t2 + cls_mid * p.Cout + n, DS_BX_T2);
            else if (p.bias) sv = DS_LD(float, p.bias + n, DS_BX_BIAS);
        }
        shift_mid[j] = sv;
    }
    T* const outp = reinterpret_cast<T*>(p.out);
    const T* const resp = reinterpret_cast<const T*>(p.res);
    const bool has_res = resp != nullptr;
    const int cout_v = (p.Cout + V - 1) / V * V;

#pragma unroll
    for (int i = 0; i < FM; ++i) {
        // ---- phase 1: fold / bias / activation in accumulator layout
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            int cls = cls_mid;
            bool border = false;
            if constexpr (NCLS9) {
                const ConvCoord c = coord(ml_base + i * 32 + row);
                cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
                border = cls != cls_mid && c.ok;
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                float sh = shift_mid[j];
                if constexpr (NCLS9) {
                    if (border && nok[j]) {
                        const int n = n_base + j * 32 + frow;
                        sh = DS_LD(float, p.fold_t1 + cls * p.Cout + n, DS_BX_T1) - gam * DS_LD(float, p.fold_t2 + cls * p.Cout + n, DS_BX_T2);
                    }
                }
                stage[row * SW + j * 32 + frow] = act_const<ACT>(ga * acc[i][j][r] + sh);
            }
        }
        // ---- phase 2: row-major stream-out, 16 bytes per lane (LDS ops of one wave execute in order)
        constexpr int CPRW = TN / V;  // 16-byte chunks per slab row
        constexpr int SLOTS = 32 * CPRW;
        static_assert(SLOTS % 64 == 0, "slab must split evenly over the wave");
#pragma unroll
        for (int it = 0; it < SLOTS / 64; ++it) {
            const int slot = lane + it * 64;
            const int row = slot / CPRW, cv = slot - row * CPRW;
            const int n = n_base + cv * V;
            const ConvCoord c = coord(ml_base + i * 32 + row);
            float v[V];
#pragma unroll
            for (int q = 0; q < V; q += 4) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + row * SW + cv * V + q);
                v[q] = t4[0]; v[q + 1] = t4[1]; v[q + 2] = t4[2]; v[q + 3] = t4[3];
            }
            if (c.ok && n < cout_v) {
                const size_t o = ((size_t)b * outHW + c.pix) * p.out_C + p.out_c0 + n;
                if (has_res) {
                    float rv[V];
                    vec16_load<T>(resp + o, rv, DS_BX_RES);
#pragma unroll
                    for (int q = 0; q < V; ++q) v[q] += rv[q];
                }
                if constexpr (!(DS_ABLATE & 32)) vec16_store<T>(outp + o, v, DS_BX_OUT);
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    s1 += v[q];
                    s2 += v[q] * v[q];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Register-only epilogue for TRANSPOSED accumulators: the kernel issues mfma(weights, pixels), so a lane holds ONE
// output pixel (lane & 31) and 16 channels 8*(r>>2) + 4*(lane>>5) + (r&3) of each 32x32 tile.  One
// v_permlane32_swap per register pairs the two lane halves' 4-channel groups into 8 consecutive channels of the
// same pixel: 16 contiguous output bytes (bf16) per lane with no LDS transpose, a per-LANE border class (no branch)
// and the per-channel shifts (bias, or GroupNorm-fold tables combined with this sample's statistics) read as
// vectors from a small LDS table the block builds once: shl[cls][BN] floats.
// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b.  Inline asm: this toolchain's __builtin_amdgcn_permlane32_swap
// is folded to a single swap when unrolled over several registers (wrong results); the s_nop covers the VALU-write ->
// permlane-read wait states the compiler would otherwise insert itself.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int BN>
__device__ __forceinline__ void conv_shift_table(const ds_conv_params& p, int n0, float gam, float* shl) {
    const bool fold = p.gn_ab != nullptr || p.gn_part != nullptr;
    const int ncls = fold ? p.ncls : 1;
    for (int e = threadIdx.x; e < ncls * BN; e += blockDim.x) {
        const int cls = e / BN, n = n0 + e - cls * BN;
        float v = 0.f;
        if (n < p.Cout) {
            if (fold) v = DS_LD(float, p.fold_t1 + cls * p.Cout + n, DS_BX_T1) - gam * DS_LD(float, p.fold_t2 + cls * p.Cout + n, DS_BX_T2);
            else if (p.bias) v = DS_LD(float, p.bias + n, DS_BX_BIAS);
        }
        shl[e] = v;
    }
}

template <typename T, int FM, int FN, int BN, int ACT, bool NCLS9, bool RAW, typename CoordFn>
__device__ __forceinline__ void conv_epilogue_t_body(const ds_conv_params& p, f32x16 (&acc)[FM][FN], int b, int n0, int n_loc, int ml_base,
                                                     int outHW, const float* shl, CoordFn coord, float& s1, float& s2, float ga) {
    constexpr int V = Vec16<T>::N;
    const int lane = threadIdx.x & 63, px = lane & 31, fh = lane >> 5;
    T* const outp = reinterpret_cast<T*>(p.out);
    const T* const resp = reinterpret_cast<const T*>(p.res);
    const bool has_res = !RAW && resp != nullptr;
    const int cout_v = (p.Cout + V - 1) / V * V;
    // residual vectors of the whole wave tile are requested up front: fetched one by one between the permute / activation
    // groups, each load's latency (a full memory round trip: the residual is a different tensor) sat in front of its store
    u32x4 rres[(!RAW && sizeof(T) == 2) ? FM * FN * 2 : 1];
    if constexpr (!RAW && sizeof(T) == 2) {
        if (has_res) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const ConvCoord c = coord(ml_base + i * 32 + px);
                const size_t obase = ((size_t)b * outHW + c.pix) * p.out_C + p.out_c0 + n0 + n_loc + 8 * fh;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int cofs = j * 32 + 16 * q;
                        const bool ok = c.ok && n0 + n_loc + 8 * fh + cofs < cout_v;
                        u32x4 v = u32x4{0u, 0u, 0u, 0u};
                        if (ok && !(DS_EPI_ABL & 2)) v = DS_LD(u32x4, resp + obase + cofs, DS_BX_RES);
                        rres[(i * FN + j) * 2 + q] = v;
                    }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");   // last MFMA write -> first (inline-asm) read of the accumulators
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const ConvCoord c = coord(ml_base + i * 32 + px);
        int cls = 0;
        if constexpr (NCLS9) cls = (c.ho == 0 ? 0 : (c.ho == p.Ho - 1 ? 2 : 1)) * 3 + (c.wo == 0 ? 0 : (c.wo == p.Wo - 1 ? 2 : 1));
        const float* shrow = shl + cls * BN + n_loc + 8 * fh;
        const size_t obase = ((size_t)b * outHW + c.pix) * p.out_C + p.out_c0 + n0 + n_loc + 8 * fh;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // lower half keeps its own group 2q and receives the upper half's group 2q; the upper half gets both of 2q+1
                    v[k] = acc[i][j][8 * q + k];
                    v[4 + k] = acc[i][j][8 * q + 4 + k];
                    permlane32_swap(v[k], v[4 + k]);
                }
                const int cofs = j * 32 + 16 * q;
                if constexpr (!RAW) {
                    const f32x4 sa = *reinterpret_cast<const f32x4*>(shrow + cofs), sb = *reinterpret_cast<const f32x4*>(shrow + cofs + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = act_const<(DS_EPI_ABL & 4) ? DS_ACT_NONE : ACT>(ga * v[k] + sa[k]);
                        v[4 + k] = act_const<(DS_EPI_ABL & 4) ? DS_ACT_NONE : ACT>(ga * v[4 + k] + sb[k]);
                    }
                }
                if (c.ok && n0 + n_loc + 8 * fh + cofs < cout_v) {
                    const size_t o = obase + cofs;
                    if (has_res) {
                        if constexpr (!RAW && sizeof(T) == 2) {
                            const u32x4 rr = rres[(i * FN + j) * 2 + q];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[2 * k] += __uint_as_float(rr[k] << 16);              // bf16 -> fp32: the low / high half of each dword
                                v[2 * k + 1] += __uint_as_float(rr[k] & 0xffff0000u);
                            }
                        } else {
#pragma unroll
                            for (int q2 = 0; q2 < 8; q2 += V) {
                                float rv[V];
                                vec16_load<T>(resp + o + q2, rv, DS_BX_RES);
#pragma unroll
                                for (int k = 0; k < V; ++k) v[q2 + k] += rv[k];
                            }
                        }
                    }
                    if constexpr (!(DS_EPI_ABL & 1)) {
#pragma unroll
                        for (int q2 = 0; q2 < 8; q2 += V) vec16_store<T>(outp + o + q2, v + q2, DS_BX_OUT);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        s1 += v[k];
                        s2 += v[k] * v[k];
                    }
                }
            }
    }
}

template <typename T, int FM, int FN, int BN, typename CoordFn>
__device__ __forceinline__ void conv_epilogue_t(const ds_conv_params& p, f32x16 (&acc)[FM][FN], int b, int n0, int n_loc, int ml_base, int outHW,
                                                const float* shl, CoordFn coord, float& s1, float& s2, float ga) {
    const bool fold = p.gn_ab != nullptr || p.gn_part != nullptr;
    if (p.act == DS_ACT_GELU) {
        if (fold && p.ncls == 9) conv_epilogue_t_body<T, FM, FN, BN, DS_ACT_GELU, true, false>(p, acc, b, n0, n_loc, ml_base, outHW, shl, coord, s1, s2, ga);
        else conv_epilogue_t_body<T, FM, FN, BN, DS_ACT_GELU, false, false>(p, acc, b, n0, n_loc, ml_base, outHW, shl, coord, s1, s2, ga);
    } else {
        if (fold && p.ncls == 9) conv_epilogue_t_body<T, FM, FN, BN, DS_ACT_NONE, true, false>(p, acc, b, n0, n_loc, ml_base, outHW, shl, coord, s1, s2, ga);
        else conv_epilogue_t_body<T, FM, FN, BN, DS_ACT_NONE, false, false>(p, acc, b, n0, n_loc, ml_base, outHW, shl, coord, s1, s2, ga);
    }
}

// U-Net convolutions use no activation or GELU in the epilogue (SiLU / ReLU of the variants run in ds_gn_apply)
template <typename T, int FM, int FN, typename CoordFn>
__device__ __forceinline__ void conv_epilogue(const ds_conv_params& p, f32x16 (&acc)[FM][FN], int b, int n_base, int ml_base,
                                              int outHW, float* stage, CoordFn coord, float& s1, float& s2, float gn_a = 1.f,
                                              float gn_am = 0.f) {
    if (p.act == DS_ACT_GELU) {
        if (p.ncls == 9) conv_epilogue_body<T, FM, FN, DS_ACT_GELU, true>(p, acc, b, n_base, ml_base, outHW, stage, coord, s1, s2, gn_a, gn_am);
        else conv_epilogue_body<T, FM, FN, DS_ACT_GELU, false>(p, acc, b, n_base, ml_base, outHW, stage, coord, s1, s2, gn_a, gn_am);
    } else {
        if (p.ncls == 9) conv_epilogue_body<T, FM, FN, DS_ACT_NONE, true>(p, acc, b, n_base, ml_base, outHW, stage, coord, s1, s2, gn_a, gn_am);
        else conv_epilogue_body<T, FM, FN, DS_ACT_NONE, false>(p, acc, b, n_base, ml_base, outHW, stage, coord, s1, s2, gn_a, gn_am);
    }
}
