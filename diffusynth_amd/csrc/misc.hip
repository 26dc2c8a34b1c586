// Small kernels around the U-Net: conditioning MLPs, boundary layout converts, the fused sampler
// step, Philox noise and the "repeat" noise column gather (gfx950).
#include <stdarg.h>

#include "common.hpp"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void ds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ds_last_error_string(void) { return g_err; }
extern "C" int ds_abi_version(void) { return 1; }

// fp32 NHWC [npix][C] -> two bf16 planes per pixel [npix][2C] (hi = bf16(x), then lo = bf16(x - hi)): the DS_CONV_F_SPLIT_IN input format
// of the split-precision convolutions, for tensors whose producer is an fp32 kernel
namespace {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* x, bf16* out, size_t nvec, int CV) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / CV;
        const int cv = (int)(i - pix * CV);
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), c = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        const float v8[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        u32x4 hi, lo;
        ds_split8(v8, hi, lo);
        bf16* o = out + (pix * 2 * CV + cv) * 8;
        *reinterpret_cast<u32x4*>(o) = hi;
        *reinterpret_cast<u32x4*>(o + (size_t)CV * 8) = lo;
    }
}
}  // namespace
extern "C" int ds_split_planes(const float* x, void* out, long long npix, int C, void* stream) {
    DS_REQUIRE(x && out && npix > 0 && C > 0 && C % 8 == 0 && ds_aligned16(x) && ds_aligned16(out), "split_planes: C %% 8 == 0, 16-byte aligned pointers");
    const size_t nvec = (size_t)npix * (C / 8);
    const int blocks = (int)((nvec + 255) / 256 < 16384 ? (nvec + 255) / 256 : 16384);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<bf16*>(out), nvec, C / 8);
    DS_CHECK_LAUNCH("split_planes");
    return DS_OK;
}

// ---- bounds diagnostics (see common.hpp).  Product build: reports "not a bounds build" (-1).
#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_igemm(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv_splitk(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv_halo3(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv_quad(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv_smalln(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_dwconv_gn(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_attn_fused(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_attn_x3(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_linattn(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv1x1_x3(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv7x7_c4(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_convt4x4_c80(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv3x3_c80(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_conv3x3_f32_n4(ds_bounds_rec*, int);
extern "C" int ds_bounds_fetch_vq_attn(ds_bounds_rec*, int);
#endif
extern "C" int ds_bounds_report(char* buf, int n, int reset) {
#if DS_BOUNDS
    static const char* knames[] = {"?", "conv_igemm", "conv3x3_halo", "splitk_reduce", "dwconv7_mfma", "dwconv7_lds", "dwconv7",
                                   "attn_fused_ctx", "attn_fused_out", "gn_apply", "linattn", "conv7x7_c4", "convt4x4_c80", "conv3x3_c80", "conv3x3_f32_n4", "vq_attn_ctx", "vq_attn_apply"};
    static const char* bnames[] = {"src0", "src1", "weights", "out", "res", "bias", "fold_t1", "fold_t2", "gn_ab", "gn_part", "stats_part",
                                   "aux0", "aux1", "aux2", "aux3"};
    int (*fetch[])(ds_bounds_rec*, int) = {ds_bounds_fetch_conv_igemm, ds_bounds_fetch_conv_splitk, ds_bounds_fetch_conv_halo3, ds_bounds_fetch_conv_quad, ds_bounds_fetch_conv_smalln, ds_bounds_fetch_dwconv_gn,
                                           ds_bounds_fetch_attn_fused, ds_bounds_fetch_attn_x3, ds_bounds_fetch_linattn, ds_bounds_fetch_conv1x1_x3, ds_bounds_fetch_conv7x7_c4, ds_bounds_fetch_convt4x4_c80, ds_bounds_fetch_conv3x3_c80, ds_bounds_fetch_conv3x3_f32_n4, ds_bounds_fetch_vq_attn};
    int hits = 0, pos = 0;
    if (buf && n > 0) buf[0] = 0;
    for (auto f : fetch) {
        ds_bounds_rec r;
        if (f(&r, reset) != 0) return -2;
        if (r.hit) {
            ++hits;
            if (buf && pos < n)
                pos += snprintf(buf + pos, n - pos, "%s: %s access of %d bytes at offset %lld outside extent %lld (block %d,%d,%d thread %d); ",
                                knames[r.kernel < 15 ? r.kernel : 0], bnames[r.buf < DS_BX_N ? r.buf : 0], r.size, r.off, r.extent, r.bx, r.by,
                                r.bz, r.tid);
        }
    }
    return hits;
#else
    if (buf && n > 0) snprintf(buf, n, "not a bounds build (compile with -DDS_BOUNDS=1: tools/build_variants.py bounds)");
    (void)reset;
    return -1;
#endif
}

namespace {

// ------------------------------------------------------------------------------------------------ conditioning
__global__ void sinusoid_kernel(const int64_t* t, const float* freqs, int B, int half, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float arg = (float)t[b] * freqs[j];
    out[(size_t)b * 2 * half + j] = sinf(arg);
    out[(size_t)b * 2 * half + half + j] = cosf(arg);
}

// one wave per output element: y[b][o] = bias[o] + sum_k act(x[b][k]) W[o][k]
__global__ __launch_bounds__(256) void linear_kernel(const float* x, int xs, const float* W, const float* bias, int B, int K,
                                                     int O, int act_in, float* y, int ys) {
    const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wid >= (long)B * O) return;
    const int b = wid / O, o = wid % O;
    const float* xr = x + (size_t)b * xs;
    const float* wr = W + (size_t)o * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(act_apply(xr[k], act_in), wr[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) y[(size_t)b * ys + o] = acc + (bias ? bias[o] : 0.f);
}

// fp32 MFMA form of the same GEMV stack (16x16x4: exact fp32 fma chains): wave = 16 outputs x NBT tiles of 16 samples.
// A lane loads 16 bytes of its x row and of its W row per 16-deep K step and feeds element s of both to MFMA s, so the k
// assignment (k0 + 4*(lane>>4) + s) agrees between the operands and W stays in nn.Linear's row-major [O][K] layout.
// No cross-lane reduction, every W element is read once per 16 samples: the stacked time-bias GEMV (O ~ 8k) drops from
// ~50 us to a few us and no longer grows with the batch.
template <int NBT>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* x, int xs, const float* W, const float* bias, int B, int K, int O,
                                                          int act_in, float* y, int ys) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int o0 = wave * 16;
    if (o0 >= O) return;                                   // wave-uniform
    const int n = lane & 15, kq = lane >> 4;
    const float* wr = W + (size_t)min(o0 + n, O - 1) * K + 4 * kq;
    const int nbt = (B + 15) / 16;
    for (int bt0 = 0; bt0 < nbt; bt0 += NBT) {
        const float* xr[NBT];
        f32x4 acc[NBT];
#pragma unroll
        for (int t = 0; t < NBT; ++t) {
            xr[t] = x + (size_t)min((bt0 + t) * 16 + n, B - 1) * xs + 4 * kq;
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 6
        for (int k0 = 0; k0 < K; k0 += 16) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k0);
#pragma unroll
            for (int t = 0; t < NBT; ++t) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(xr[t] + k0);
                if (act_in != DS_ACT_NONE) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) xv[s4] = act_apply(xv[s4], act_in);
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s4], wv[s4], acc[t], 0, 0, 0);
            }
        }
        const int o = o0 + n;
        const float bo = (bias && o < O) ? bias[o] : 0.f;
#pragma unroll
        for (int t = 0; t < NBT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = (bt0 + t) * 16 + kq * 4 + r;
                if (b < B && o < O) y[(size_t)b * ys + o] = acc[t][r] + bo;
            }
    }
}

// ------------------------------------------------------------------------------------------------ layouts
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, int C, int HW, T* out, int Cp, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % Cp;
        const size_t bp = i / Cp;
        const size_t b = bp / HW, pix = bp % HW;
        out[i] = from_f32<T>(c < C ? x[(b * C + c) * HW + pix] : 0.f);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, int C, int Cs, int HW, float* out, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % HW;
        const size_t bc = i / HW;
        const size_t b = bc / C, c = bc % C;
        out[i] = to_f32(x[(b * HW + pix) * Cs + c]);
    }
}

// ------------------------------------------------------------------------------------------------ sampler step
// Every operation below is a separately rounded IEEE fp32 op in the reference's order
// (DiffSynthSampler.py:320,327,337,343,291-293,506): contraction into FMA is switched off.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void ddim_step_kernel(const ds_step_params p, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = i / p.CHW;
        const float* cf = p.coef + (size_t)b * 5;
        float eps = p.eps[i];
        if (p.eps_cond) {
            const float d = p.eps_cond[i] - eps;
            const float sd = p.cfg_scale * d;
            eps = eps + sd;
        }
        const float x = p.x[i];
        const float t0 = cf[0] * eps;
        const float t1 = x - t0;
        const float x0 = t1 / cf[1];
        const float u0 = cf[2] * x0;
        const float u1 = cf[3] * eps;
        const float u2 = cf[4] * p.noise[i];
        float v = (u0 + u1) + u2;
        if (p.blend_mode) {
            const size_t r = i - (size_t)b * p.CHW;
            const float m = p.mask_chw ? p.mask[i] : p.mask[(size_t)b * p.HW + r % p.HW];
            float g = p.guide[i];
            if (p.blend_mode == 1) {
                const float* qc = p.qcoef + (size_t)b * 2;
                const float g0 = qc[0] * g;
                const float g1 = qc[1] * p.init_noise[i];
                g = g0 + g1;
            }
            const float w0 = m * g;
            const float w1 = (1.0f - m) * v;
            v = w0 + w1;
        }
        p.out[i] = v;
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}
__global__ void philox_normal_kernel(float* out, size_t n, uint64_t seed, uint64_t offset) {
    const size_t nq = (n + 3) / 4;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            float s, cs;
            sincospif(2.0f * u2, &s, &cs);
            z[2 * h] = r * cs;
            z[2 * h + 1] = r * s;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = z[j];
    }
}

__global__ void gather_cols_kernel(const float* src, int src_w, const int32_t* cols, int out_w, float* out, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / out_w;
        const int j = i % out_w;
        out[i] = src[row * src_w + cols[j]];
    }
}

inline int blocks_for(size_t n, int cap = 8192) {
    const size_t b = (n + 255) / 256;
    return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

}  // namespace

namespace {
// one block per row: two-pass (mean, then centred variance) in fp32 with double partial sums
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* a, const float* r, const float* gamma, const float* beta, int D,
                                                            float eps, float* out) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* ab = a + (size_t)b * D;
    const float* rb = r + (size_t)b * D;
    double s = 0.0;
    for (int i = tid; i < D; i += 256) s += (double)(ab[i] + rb[i]);
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float mean = (float)(red[0] / D);
    __syncthreads();
    double q = 0.0;
    for (int i = tid; i < D; i += 256) {
        const float d = (ab[i] + rb[i]) - mean;
        q += (double)d * d;
    }
    red[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float rstd = 1.0f / sqrtf((float)(red[0] / D) + eps);
    for (int i = tid; i < D; i += 256) out[(size_t)b * D + i] = ((ab[i] + rb[i]) - mean) * rstd * gamma[i] + beta[i];
}
}  // namespace

extern "C" int ds_add_layernorm(const float* a, const float* r, const float* gamma, const float* beta, int B, int D, float eps, float* out,
                                void* stream) {
    DS_REQUIRE(a && r && gamma && beta && out && B > 0 && D > 0, "add_layernorm: bad args");
    hipLaunchKernelGGL(add_layernorm_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, r, gamma, beta, D, eps, out);
    DS_CHECK_LAUNCH("add_layernorm");
    return DS_OK;
}

extern "C" int ds_sinusoid(const int64_t* t, const float* freqs, int B, int half, float* out, void* stream) {
    DS_REQUIRE(t && freqs && out && B > 0 && half > 0, "sinusoid: bad args");
    hipLaunchKernelGGL(sinusoid_kernel, dim3((B * half + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t, freqs, B, half, out);
    DS_CHECK_LAUNCH("sinusoid");
    return DS_OK;
}

// y = act(x) elementwise (fp32; may run in place).  ds_linear's act_in evaluates the activation once per (16-output wave, element): for a
// wide stack of outputs over one small input (the 44 per-block time biases from the 128 x 384 time embedding: 840 waves x 49k exact-erf GELUs,
// 85 us) the host applies it once with this kernel and calls ds_linear with DS_ACT_NONE (same fp32 function: identical results).
__global__ __launch_bounds__(256) void act_kernel(const float* x, size_t n, int act, float* y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = act_apply(x[i], act);
}

extern "C" int ds_activation(const float* x, size_t n, int act, float* y, void* stream) {
    DS_REQUIRE(x && y && n > 0, "activation: bad args");
    DS_REQUIRE(act == DS_ACT_NONE || act == DS_ACT_GELU || act == DS_ACT_SILU, "activation: unknown act %d", act);
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, act, y);
    DS_CHECK_LAUNCH("activation");
    return DS_OK;
}

extern "C" int ds_linear(const float* x, int xs, const float* W, const float* bias, int B, int K, int O, int act_in, float* y,
                         int ys, void* stream) {
    DS_REQUIRE(x && W && y && B > 0 && K > 0 && O > 0 && xs >= K && ys >= O, "linear: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (K % 16 == 0 && xs % 4 == 0 && ds_aligned16(x) && ds_aligned16(W)) {
        const int nw = (O + 15) / 16;
        dim3 grid((nw + 3) / 4), blk(256);
        if (B <= 16) hipLaunchKernelGGL(linear_mfma_kernel<1>, grid, blk, 0, st, x, xs, W, bias, B, K, O, act_in, y, ys);
        else if (B <= 32) hipLaunchKernelGGL(linear_mfma_kernel<2>, grid, blk, 0, st, x, xs, W, bias, B, K, O, act_in, y, ys);
        else if (B <= 64) hipLaunchKernelGGL(linear_mfma_kernel<4>, grid, blk, 0, st, x, xs, W, bias, B, K, O, act_in, y, ys);
        else hipLaunchKernelGGL(linear_mfma_kernel<8>, grid, blk, 0, st, x, xs, W, bias, B, K, O, act_in, y, ys);
        DS_CHECK_LAUNCH("linear_mfma");
        return DS_OK;
    }
    const long waves = (long)B * O;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, xs, W,
                       bias, B, K, O, act_in, y, ys);
    DS_CHECK_LAUNCH("linear");
    return DS_OK;
}

extern "C" int ds_nchw_to_nhwc(const float* x, int B, int C, int H, int W, void* out, int Cp, int dtype, void* stream) {
    DS_REQUIRE(x && out && B > 0 && C > 0 && Cp >= C && H > 0 && W > 0, "nchw_to_nhwc: bad args");
    const size_t total = (size_t)B * H * W * Cp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DS_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(blocks_for(total)), dim3(256), 0, st, x, C, H * W, (bf16*)out, Cp, total);
    else if (dtype == DS_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, st, x, C, H * W, (float*)out, Cp, total);
    else DS_FAIL(DS_EINVAL, "nchw_to_nhwc: dtype %d", dtype);
    DS_CHECK_LAUNCH("nchw_to_nhwc");
    return DS_OK;
}

namespace {
// First layer of the VQGAN decoder (VQGAN.py:345: Conv2d(embedding_dim, hidden, 1, bias=False) on the NCHW latent): out[b][pix][co] = sum_ci w[co][ci] x[b][ci][pix],
// bf16 NHWC.  The layout change + the generic 1x1 tile (K padded to 32, N to 96) were 11 + 85 us for a 168 MB output; here a thread takes one
// pixel's Cin values (coalesced along the pixels of each plane) and writes one 16-byte piece of its output row.
template <int CIN>
__global__ __launch_bounds__(256) void conv1x1_in_nchw_kernel(const float* x, const float* w, const float* bias, int HW, int Cout, bf16* out, size_t npix) {
    // blockDim = NP x rows: a thread keeps ONE 16-byte output piece (its 8 x CIN weights in registers) and walks pixels (the first form fetched
    // its 32 weights per output piece: 160 us, slower than what it replaced)
    const int NP = Cout >> 3, rows = blockDim.x / NP, piece = threadIdx.x % NP, row = threadIdx.x / NP;
    float wr[8][CIN], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bv[j] = bias ? bias[piece * 8 + j] : 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c) wr[j][c] = w[(piece * 8 + j) * CIN + c];
    }
    for (size_t bp = (size_t)blockIdx.x * rows + row; bp < npix; bp += (size_t)gridDim.x * rows) {
        const size_t b = bp / HW, pix = bp % HW;
        float xv[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) xv[c] = x[(b * CIN + c) * HW + pix];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = bv[j];
#pragma unroll
            for (int c = 0; c < CIN; ++c) a = fmaf(wr[j][c], xv[c], a);
            o[j] = a;
        }
        Vec16<bf16>::store(out + (bp * Cout + piece * 8), o);
    }
}
}  // namespace

extern "C" int ds_conv1x1_in_nchw(const float* x, int B, int Cin, int HW, const float* w, const float* bias, int Cout, void* out, void* stream) {
    DS_REQUIRE(x && w && out && B > 0 && HW > 0 && Cout > 0 && Cout % 8 == 0, "conv1x1_in_nchw: bad args (Cout %% 8 == 0)");
    DS_REQUIRE(Cin == 4 || Cin == 8, "conv1x1_in_nchw: Cin=%d unsupported (4, 8)", Cin);
    if (!ds_aligned16(out)) DS_FAIL(DS_EALIGN, "conv1x1_in_nchw: out must be 16-byte aligned");
    DS_REQUIRE(Cout / 8 <= 256, "conv1x1_in_nchw: Cout=%d too large", Cout);
    const size_t npix = (size_t)B * HW;
    const int NP = Cout / 8, threads = NP * (256 / NP), rows = threads / NP;
    const int blocks = blocks_for((npix + rows - 1) / rows * 64, 16384);                 // (blocks_for counts 256 items per block: >= 4 pixels per thread)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (Cin == 4) hipLaunchKernelGGL(conv1x1_in_nchw_kernel<4>, dim3(blocks), dim3(threads), 0, st, x, w, bias, HW, Cout, (bf16*)out, npix);
    else hipLaunchKernelGGL(conv1x1_in_nchw_kernel<8>, dim3(blocks), dim3(threads), 0, st, x, w, bias, HW, Cout, (bf16*)out, npix);
    DS_CHECK_LAUNCH("conv1x1_in_nchw");
    return DS_OK;
}

namespace {
// dst[0, n) = dst[n, 2n) = src[0, n) in 16-byte pieces: one read, two writes (two device-to-device copies read the source twice)
// INPLACE (dst == src, r05): the first half is already where it belongs — one read, ONE write (the plan computes the shared prefix of a
// classifier-free-guidance batch straight into the first half of the full-batch tensor)
template <bool INPLACE>
__global__ __launch_bounds__(256) void dup_batch_kernel(const u32x4* src, u32x4* dst, size_t nvec) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = src[i];
        if (!INPLACE) dst[i] = v;
        dst[i + nvec] = v;
    }
}
}  // namespace

extern "C" int ds_dup_batch(const void* src, void* dst, size_t nbytes, void* stream) {
    DS_REQUIRE(src && dst && nbytes > 0, "dup_batch: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nbytes % 16 == 0 && ds_aligned16(src) && ds_aligned16(dst)) {
        const size_t nvec = nbytes / 16;
        const size_t want = (nvec + 255) / 256;
        const dim3 grid((unsigned)(want < 16384 ? want : 16384));
        if (src == dst) hipLaunchKernelGGL(dup_batch_kernel<true>, grid, dim3(256), 0, st, reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(dst), nvec);
        else hipLaunchKernelGGL(dup_batch_kernel<false>, grid, dim3(256), 0, st, reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(dst), nvec);
        DS_CHECK_LAUNCH("dup_batch");
        return DS_OK;
    }
    hipError_t e = src == dst ? hipSuccess : hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(static_cast<char*>(dst) + nbytes, src, nbytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) DS_FAIL(DS_ELAUNCH, "dup_batch: %s", hipGetErrorString(e));
    return DS_OK;
}

extern "C" int ds_nhwc_to_nchw(const void* x, int dtype, int B, int C, int Cs, int H, int W, float* out, void* stream) {
    DS_REQUIRE(x && out && B > 0 && C > 0 && Cs >= C && H > 0 && W > 0, "nhwc_to_nchw: bad args");
    const size_t total = (size_t)B * C * H * W;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DS_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(blocks_for(total)), dim3(256), 0, st, (const bf16*)x, C, Cs, H * W, out, total);
    else if (dtype == DS_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, st, (const float*)x, C, Cs, H * W, out, total);
    else DS_FAIL(DS_EINVAL, "nhwc_to_nchw: dtype %d", dtype);
    DS_CHECK_LAUNCH("nhwc_to_nchw");
    return DS_OK;
}

extern "C" int ds_ddim_step(const ds_step_params* p, void* stream) {
    DS_REQUIRE(p && p->x && p->eps && p->noise && p->out && p->coef, "ddim_step: null pointer");
    DS_REQUIRE(p->B > 0 && p->CHW > 0 && p->HW > 0 && p->CHW % p->HW == 0, "ddim_step: bad sizes");
    DS_REQUIRE(p->blend_mode >= 0 && p->blend_mode <= 2, "ddim_step: blend_mode %d", p->blend_mode);
    DS_REQUIRE(p->blend_mode == 0 || (p->guide && p->mask), "ddim_step: blend needs guide and mask");
    DS_REQUIRE(p->blend_mode != 1 || (p->init_noise && p->qcoef), "ddim_step: blend 1 needs init_noise and qcoef");
    const size_t total = (size_t)p->B * p->CHW;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(blocks_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *p, total);
    DS_CHECK_LAUNCH("ddim_step");
    return DS_OK;
}

extern "C" int ds_philox_normal(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
    DS_REQUIRE(out && n > 0, "philox_normal: bad args");
    hipLaunchKernelGGL(philox_normal_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, n, seed, offset);
    DS_CHECK_LAUNCH("philox_normal");
    return DS_OK;
}

extern "C" int ds_gather_cols(const float* src, int rows, int src_w, const int32_t* cols, int out_w, float* out, void* stream) {
    DS_REQUIRE(src && cols && out && rows > 0 && src_w > 0 && out_w > 0, "gather_cols: bad args");
    const size_t total = (size_t)rows * out_w;
    hipLaunchKernelGGL(gather_cols_kernel, dim3(blocks_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, src_w, cols, out_w, out, total);
    DS_CHECK_LAUNCH("gather_cols");
    return DS_OK;
}
