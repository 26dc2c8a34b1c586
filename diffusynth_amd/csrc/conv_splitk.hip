// Split-K of the convolutions at small batch (too few blocks for 256 CUs): the K slices of ds_conv_igemm (generic kernel) and of
// conv3x3_halo3 store raw fp32 partial sums to p.slab[z][B][pixel][roundup(Cout, 8)]; this kernel adds the slices and runs the epilogue
// (GroupNorm fold or bias, activation, residual, statistics partials) — the bf16 tier, and (r04) the split-precision launches of the bf16x3
// tier: its 3x3 kernel has three times the K steps per block, and at small batches a 32 x 8-level layer was 8 blocks of 648 serial steps
// (the all-fp32 tier never splits: batch-invariant results).
#include "common.hpp"
#include "conv_epilogue.hpp"
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif

namespace {

// ---- split-K reduce + epilogue: one thread per 8 output channels of one pixel
constexpr int RED_BLOCK = 256;
template <int KS>
__global__ __launch_bounds__(RED_BLOCK) void splitk_reduce_kernel(const ds_conv_params p) {
    __shared__ float red[2 * (RED_BLOCK / 64)];
    const int b = blockIdx.y;
    const int oH = p.transposed ? 2 * p.Ho : p.Ho, oW = p.transposed ? 2 * p.Wo : p.Wo;   // output image (= input for the 3x3 halo path)
    const int HW = oH * oW, cv8 = (p.Cout + 7) / 8, cs = cv8 * 8;
    const long i = (long)blockIdx.x * RED_BLOCK + threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    const bool live = i < (long)HW * cv8;
    const int pix = live ? (int)(i / cv8) : 0, n = live ? (int)((i - (long)pix * cv8) * 8) : 0;
    // all KS slab reads in flight at once (with a rolled loop each pair of loads waited for the previous one), and requested BEFORE the
    // statistics of the input are reduced: at a small batch this kernel is a chain of round trips, 7 - 11 us for a few hundred KB (r04)
    f32x4 sa[KS], sc[KS];
#pragma unroll
    for (int z = 0; z < KS; ++z) {
        const float* sp = p.slab + (((size_t)z * p.B + b) * HW + pix) * cs + n;
        sa[z] = DS_LD(f32x4, sp, DS_BX_AUX0);
        sc[z] = DS_LD(f32x4, sp + 4, DS_BX_AUX0);
    }
    // the statistics reduction is a WAVE collective (every lane contributes partials): all threads run it, also those of a ragged last wave
    float ga = 1.f, gam = 0.f;
    const bool fold = p.gn_ab || p.gn_part;
    if (fold) {
        if (p.gn_part) {
            gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, ga, gam);
        } else {
            ga = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
            gam = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
        }
    }
    if (live) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int z = 0; z < KS; ++z) {
            v[0] += sa[z][0]; v[1] += sa[z][1]; v[2] += sa[z][2]; v[3] += sa[z][3];
            v[4] += sc[z][0]; v[5] += sc[z][1]; v[6] += sc[z][2]; v[7] += sc[z][3];
        }
        int cls = 0;
        if (fold) {
            if (p.ncls == 9) {
                const int ho = pix / oW, wo = pix - ho * oW;
                cls = (ho == 0 ? 0 : (ho == oH - 1 ? 2 : 1)) * 3 + (wo == 0 ? 0 : (wo == oW - 1 ? 2 : 1));
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float sh = 0.f;
            if (n + q < p.Cout) {
                if (fold) sh = DS_LD(float, p.fold_t1 + cls * p.Cout + n + q, DS_BX_T1) - gam * DS_LD(float, p.fold_t2 + cls * p.Cout + n + q, DS_BX_T2);
                else if (p.bias) sh = DS_LD(float, p.bias + n + q, DS_BX_BIAS);
            }
            v[q] = ga * v[q] + sh;
            if (p.act == DS_ACT_GELU) v[q] = gelu_fast(v[q]);
        }
        const int out_mode = (p.flags >> 1) & 3;                     // DS_CONV_F_*: 1 = hi / lo bf16 planes (exact-erf GELU upstream), 2 = fp32 (+ fp32 residual)
        if (out_mode == 1) {
            if (n < p.Cout) {
                bf16* o2 = reinterpret_cast<bf16*>(p.out) + ((size_t)b * HW + pix) * p.out_C + p.out_c0 + n;
                u32x4 hi, lo;
                ds_split8(v, hi, lo);
                DS_ST(u32x4, reinterpret_cast<u32x4*>(o2), DS_BX_OUT, hi);
                DS_ST(u32x4, reinterpret_cast<u32x4*>(o2 + p.Cout), DS_BX_OUT, lo);
            }
        } else if (out_mode == 2) {
            float* of = reinterpret_cast<float*>(p.out) + ((size_t)b * HW + pix) * p.out_C + p.out_c0 + n;
            if (p.res) {
                const float* rf = reinterpret_cast<const float*>(p.res) + ((size_t)b * HW + pix) * p.out_C + p.out_c0 + n;
                const f32x4 r0 = DS_LD(f32x4, rf, DS_BX_RES), r1 = DS_LD(f32x4, rf + 4, DS_BX_RES);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] += r0[q];
                    v[4 + q] += r1[q];
                }
            }
            DS_ST(f32x4, reinterpret_cast<f32x4*>(of), DS_BX_OUT, (f32x4{v[0], v[1], v[2], v[3]}));
            DS_ST(f32x4, reinterpret_cast<f32x4*>(of + 4), DS_BX_OUT, (f32x4{v[4], v[5], v[6], v[7]}));
        } else {
        const size_t o = ((size_t)b * HW + pix) * p.out_C + p.out_c0 + n;
        if (p.res) {
            float rv[8];
            vec16_load<bf16>(reinterpret_cast<const bf16*>(p.res) + o, rv, DS_BX_RES);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += rv[q];
        }
        vec16_store<bf16>(reinterpret_cast<bf16*>(p.out) + o, v, DS_BX_OUT);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s1 += v[q];
            s2 += v[q] * v[q];
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

}  // namespace

extern "C" int ds_conv_splitk_reduce(const ds_conv_params* p, void* stream) {
    DS_REQUIRE(p && (p->ksplit == 2 || p->ksplit == 3 || p->ksplit == 4 || p->ksplit == 6 || p->ksplit == 8) && p->slab && p->out,
               "splitk_reduce: needs ksplit in {2, 3, 4, 6, 8}, slab and out");
    DS_REQUIRE(p->dtype == DS_BF16, "splitk_reduce: bf16 only");
    const long nvec = (long)p->Ho * p->Wo * (p->transposed ? 4 : 1) * ((p->Cout + 7) / 8);
    dim3 grid((unsigned)((nvec + RED_BLOCK - 1) / RED_BLOCK), p->B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        ds_conv_params q = *p;
        q.ksplit = 1;                          // OUT = the real output; the slab is this kernel's AUX0 input
        DsBxHost h(DS_K_SPLITK_REDUCE);
        ds_conv_bounds_table(q, DS_K_SPLITK_REDUCE, grid.x, &h.t);
        const long long oHW = (long long)p->Ho * p->Wo * (p->transposed ? 4 : 1);
        h.set(DS_BX_AUX0, p->slab, (long long)p->ksplit * p->B * oHW * ((p->Cout + 7) / 8 * 8) * 4);
        if (((p->flags >> 1) & 3) == 2) {                     // fp32 output / residual: twice the bytes the bf16 description implies
            h.set(DS_BX_OUT, p->out, (long long)p->B * oHW * p->out_C * 4);
            h.set(DS_BX_RES, p->res, (long long)p->B * oHW * p->out_C * 4);
        }
        h.publish(st);
    }
#endif
    if (p->ksplit == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, grid, dim3(RED_BLOCK), 0, st, *p);
    else if (p->ksplit == 3) hipLaunchKernelGGL(splitk_reduce_kernel<3>, grid, dim3(RED_BLOCK), 0, st, *p);      // (r04: channel counts are multiples of
    else if (p->ksplit == 6) hipLaunchKernelGGL(splitk_reduce_kernel<6>, grid, dim3(RED_BLOCK), 0, st, *p);      // 96: chunk counts of 3, 9, 18 only split by 3s)
    else if (p->ksplit == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(RED_BLOCK), 0, st, *p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<8>, grid, dim3(RED_BLOCK), 0, st, *p);
    DS_CHECK_LAUNCH("splitk_reduce");
    return DS_OK;
}

// statistics partials of a split-K launch = one per block of the reduce kernel
int ds_conv_splitk_parts(const ds_conv_params* p) {
    return (int)(((long)p->Ho * p->Wo * (p->transposed ? 4 : 1) * ((p->Cout + 7) / 8) + RED_BLOCK - 1) / RED_BLOCK);
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_splitk(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
