// Shared device/host helpers for libdiffusynth_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/diffusynth_hip.h"

typedef __bf16 bf16;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local message, negative return codes) -------------------------------
void ds_set_error(const char* fmt, ...);
#define DS_FAIL(code, ...)        \
    do {                          \
        ds_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)
#define DS_REQUIRE(cond, ...) \
    do {                      \
        if (!(cond)) DS_FAIL(DS_EINVAL, __VA_ARGS__); \
    } while (0)
#define DS_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) DS_FAIL(DS_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline bool ds_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: a launcher remembers, per kernel
// instantiation, the devices it has already set it on (bit d of the mask; idempotent, so a race is benign).
#include <atomic>
struct DsDevOnce {
    std::atomic<unsigned long long> mask{0};
    // true when the CURRENT device still needs the attribute; *dev receives the device ordinal
    bool need(int* dev) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        *dev = d;
        return d >= 64 || !((mask.load(std::memory_order_relaxed) >> d) & 1ull);
    }
    void done(int dev) {
        if (dev < 64) mask.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
};
#define DS_SET_MAX_LDS(kern, bytes, name)                                                                          \
    do {                                                                                                           \
        static DsDevOnce once_;                                                                                    \
        int dev_;                                                                                                  \
        if (once_.need(&dev_)) {                                                                                   \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) DS_FAIL(DS_ELAUNCH, "%s: hipFuncSetAttribute(%d): %s", name, (int)(bytes), hipGetErrorString(e_)); \
            once_.done(dev_);                                                                                      \
        }                                                                                                          \
    } while (0)

// ---- diagnostic bounds build (-DDS_BOUNDS=1; tools/build_variants.py bounds -> libdiffusynth_hip_bounds.so) ------
// Every global access of the convolution / depthwise / attention kernels that is unconditional, clamped or guarded
// by tile arithmetic goes through DS_LD / DS_ST.  In the product build these are plain vector accesses.  In the
// bounds build the launcher computes the byte extent of each operand FROM THE PARAMETER STRUCT (what the caller
// promised), publishes it in a per-translation-unit __device__ table ahead of the launch (stream-ordered copy), and
// each access first checks [ptr, ptr + size) against its operand's extent: a violation is recorded (first one wins:
// kernel id, operand, block, thread, byte offset, extent) and the access is suppressed (loads return zero), so the
// diagnostic run itself can never fault.  ds_bounds_report() collects the records of all translation units.
#ifndef DS_BOUNDS
#define DS_BOUNDS 0
#endif
// blocks per CU the hand-scheduled convolution kernels are compiled for (__launch_bounds__): two in the product build (256 VGPRs, no
// scratch: tests/test_isa_schedule_cpu.py); the checker's extra live state does not fit that budget, so the diagnostic build takes one
// block per CU (512 VGPRs) — DESIGN §4c has what a spilling build of this code did and why.  -DDS_MINBLK=2 reproduces that build.
#ifndef DS_MINBLK
#define DS_MINBLK (DS_BOUNDS ? 1 : 2)
#endif
enum { DS_BX_SRC0 = 0, DS_BX_SRC1, DS_BX_W, DS_BX_OUT, DS_BX_RES, DS_BX_BIAS, DS_BX_T1, DS_BX_T2, DS_BX_GNAB, DS_BX_GNPART,
       DS_BX_STATS, DS_BX_AUX0, DS_BX_AUX1, DS_BX_AUX2, DS_BX_AUX3, DS_BX_N };
struct ds_bx {
    const char* lo[DS_BX_N];
    long long bytes[DS_BX_N];
    int kernel;
};
struct ds_bounds_rec {
    unsigned int hit;
    int kernel, buf, tid, bx, by, bz, size;
    long long off, extent;
};
#if DS_BOUNDS
static __device__ ds_bx g_ds_bx;
static __device__ ds_bounds_rec g_ds_rec;
#ifndef DS_BOUNDS_NOCHECK
#define DS_BOUNDS_NOCHECK 0     // debugging the checker itself: 1 = every access passes without looking at the table
#endif
__device__ __forceinline__ bool ds_bx_ok(const void* ptr, int buf, int size) {
    if (DS_BOUNDS_NOCHECK) return true;
#ifdef DS_BX_SKIP_MASK
    if ((DS_BX_SKIP_MASK >> buf) & 1) return true;
#endif
    if (g_ds_bx.kernel == 0) return true;            // launcher without an extent table: unchecked
    const long long off = reinterpret_cast<const char*>(ptr) - g_ds_bx.lo[buf];
    if (off >= 0 && off + size <= g_ds_bx.bytes[buf]) return true;
    if (atomicCAS(&g_ds_rec.hit, 0u, 1u) == 0u) {
        g_ds_rec.kernel = g_ds_bx.kernel; g_ds_rec.buf = buf; g_ds_rec.tid = threadIdx.x;
        g_ds_rec.bx = blockIdx.x; g_ds_rec.by = blockIdx.y; g_ds_rec.bz = blockIdx.z; g_ds_rec.size = size;
        g_ds_rec.off = off; g_ds_rec.extent = g_ds_bx.bytes[buf];
    }
    return false;
}
template <typename V> __device__ __forceinline__ V ds_ld_checked(const void* ptr, int buf) {
    V z;
    __builtin_memset(&z, 0, sizeof(V));
    return ds_bx_ok(ptr, buf, (int)sizeof(V)) ? *reinterpret_cast<const V*>(ptr) : z;
}
template <typename V> __device__ __forceinline__ void ds_st_checked(void* ptr, int buf, const V& v) {
    if (ds_bx_ok(ptr, buf, (int)sizeof(V))) *reinterpret_cast<V*>(ptr) = v;
}
#define DS_LD(V, ptr, buf) ds_ld_checked<V>((ptr), (buf))
#define DS_ST(V, ptr, buf, val) ds_st_checked<V>((ptr), (buf), (val))
// host side: publish the extents for the next launch on `st`
struct DsBxHost {
    ds_bx t;
    explicit DsBxHost(int kernel) { memset(&t, 0, sizeof(t)); t.kernel = kernel; }
    DsBxHost& set(int buf, const void* base, long long bytes) {
        t.lo[buf] = reinterpret_cast<const char*>(base);
        t.bytes[buf] = base ? bytes : 0;
        return *this;
    }
    // (synchronous on purpose: the table lives on the caller's stack and is shared by every launch of the translation unit — the previous
    // launch must have finished with it, and an asynchronous copy could read the stack after it has been reused.  Diagnostic build only.)
    void publish(hipStream_t st) {
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ds_bx), &t, sizeof(t), 0, hipMemcpyHostToDevice);
    }
};
static inline int ds_bounds_fetch_tu(ds_bounds_rec* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ds_rec), sizeof(*out)) != hipSuccess) return -1;
    if (reset) {
        ds_bounds_rec z;
        memset(&z, 0, sizeof(z));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ds_rec), &z, sizeof(z));
    }
    return 0;
}
#else
#define DS_LD(V, ptr, buf) (*reinterpret_cast<const V*>(ptr))
#define DS_ST(V, ptr, buf, val) (*reinterpret_cast<V*>(ptr) = (val))
#endif
// kernel ids of the bounds records
enum { DS_K_CONV_IGEMM = 1, DS_K_CONV_HALO, DS_K_SPLITK_REDUCE, DS_K_DWCONV_MFMA, DS_K_DWCONV_LDS, DS_K_DWCONV, DS_K_ATTN_CTX,
       DS_K_ATTN_OUT, DS_K_GN_APPLY, DS_K_LINATTN, DS_K_CONV7X7_C4, DS_K_CONVT4X4_C80, DS_K_CONV3X3_C80, DS_K_CONV3X3_F32_N4, DS_K_VQ_ATTN_CTX, DS_K_VQ_ATTN_APPLY };

// ---- element traits ---------------------------------------------------------------------------------
template <typename T> struct ElemTr;
template <> struct ElemTr<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int DT = DS_F32;
};
template <> struct ElemTr<bf16> {
    static constexpr int EPC = 8;
    static constexpr int DT = DS_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector of T <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float* f) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = v[i];
    }
    __device__ static __forceinline__ void store(float* p, const float* f) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f[i];
        *reinterpret_cast<f32x4*>(p) = v;
    }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const bf16* p, float* f) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    __device__ static __forceinline__ void store(bf16* p, const float* f) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};

// checked forms of Vec16<T>::load / store (plain in the product build)
template <typename T> __device__ __forceinline__ void vec16_load(const T* p, float* f, int buf) {
#if DS_BOUNDS
    if (!ds_bx_ok(p, buf, 16)) {
#pragma unroll
        for (int i = 0; i < Vec16<T>::N; ++i) f[i] = 0.f;
        return;
    }
#endif
    Vec16<T>::load(p, f);
}
template <typename T> __device__ __forceinline__ void vec16_store(T* p, const float* f, int buf) {
#if DS_BOUNDS
    if (!ds_bx_ok(p, buf, 16)) return;
#endif
    Vec16<T>::store(p, f);
}

// ---- split precision: one fp32 operand = hi + lo bf16 operands (x_hi = bf16(x), x_lo = bf16(x - x_hi)) -------------------------------------
// Pairs: one v_cvt_pk_bf16_f32 per TWO values (a scalar `(bf16)x` spends a whole v_cvt_pk per value and a v_perm / v_and_or to pack it), the
// hi halves back to fp32 by a shift / a mask of the packed dword: 3 VALU per value instead of ~4.5.
typedef float ds_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ds_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ds_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const ds_f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, ds_bf16x2));
    const ds_f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ds_bf16x2));
}
// v[0..7] -> two packed 16-byte operands
__device__ __forceinline__ void ds_split8(const float* v, u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ds_split2(v[2 * j], v[2 * j + 1], h[j], l[j]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

// ---- activations (erf GELU like nn.GELU(); x*sigmoid(x) like nn.SiLU / VQGAN swish) ----------------
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 rounding level): one v_rcp, one v_exp
// and five fma instead of libm erff's ~40 instructions — the GELU epilogue of the 3x3 convolutions
// evaluates it 49k times per 256x192 tile, which made the epilogue as long as the whole K loop.
__device__ __forceinline__ float gelu_fast(float v) {
    // 0.5 v (1 + erf(v / sqrt 2)) with erf(|z|) = 1 - q, q = poly(t) t exp(-z^2), t = 1 / (1 + 0.3275911 |z|):
    //   v >= 0: v - 0.5 v q,  v < 0: 0.5 v q   =>   relu(v) - 0.5 |v| q.
    // v_rcp_f32 / v_exp_f32 directly: the correctly rounded __frcp_rn expands to the 11-instruction IEEE division
    // sequence, which was half of the whole activation.
    const float av = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, av, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(v * v * (-0.5f * 1.44269504088896340736f));   // exp(-v^2 / 2)
    return fmaf(-0.5f * av, poly * t * e, fmaxf(v, 0.0f));
}
__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case DS_ACT_GELU: return gelu_fast(v);
        case DS_ACT_SILU: return v / (1.0f + expf(-v));
        case DS_ACT_RELU: return fmaxf(v, 0.0f);
        default: return v;
    }
}

// ---- reductions ---------------------------------------------------------------------------------------
// Wave-wide sums on the DPP network: four cross-lane adds inside each row of 16 lanes (quad swaps, half-row and row mirrors), then
// the four row totals through v_readlane.  The __shfl_xor butterfly lowers to six dependent ds_bpermute_b32 (an LDS crossbar round
// trip each, ~0.1 us): the two-sum epilogue reduction of every convolution / depthwise block cost 1.2 us, the float64 pair in
// every prologue (gn_from_partials) as much again.  Fixed summation order: deterministic.
#ifndef DS_NO_DPP_REDUCE
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);           // row_half_mirror
    v += dpp_f32<0x140>(v);           // row_mirror: every lane holds its row's sum
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const long b = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long b = __builtin_bit_cast(long, v);
    return __builtin_bit_cast(double, ((long)__builtin_amdgcn_readlane((int)(b >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)b, l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block (sum, sumsq) -> one partial pair written by thread 0.  red must hold 2*(blockDim/64) floats.
// (lane / wave given by the caller: a kernel at its register budget keeps the wave index in an SGPR and takes the lane from mbcnt instead of
// holding threadIdx.x in a VGPR across its K loop)
__device__ __forceinline__ void block_stats_write(float s1, float s2, float* red, float* dst, int lane, int wave) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int nw = blockDim.x >> 6;
    if (lane == 0) {
        red[2 * wave] = s1;
        red[2 * wave + 1] = s2;
    }
    __syncthreads();
    if (lane == 0 && wave == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < nw; ++w) {
            a += red[2 * w];
            b += red[2 * w + 1];
        }
        DS_ST(float, dst, DS_BX_STATS, a);
        DS_ST(float, dst + 1, DS_BX_STATS, b);
    }
}
__device__ __forceinline__ void block_stats_write(float s1, float s2, float* red, float* dst) {
    block_stats_write(s1, s2, red, dst, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6));
}

// Two-phase form of gn_from_partials for prologues that have other loads to issue: gn_partials_issue() requests up to
// 4 x 64 partial pairs (8-byte loads, no wait), gn_partials_finish() reduces them in float64 (and walks any remaining ones).
struct GnPartialLoads { float2 v[4]; };
__device__ __forceinline__ void gn_partials_issue(const float* part, int parts, int b, GnPartialLoads& g) {
    const int lane = threadIdx.x & 63;
    const float2* pp = reinterpret_cast<const float2*>(part + (size_t)b * parts * 2);
#if DS_BOUNDS
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        g.v[k] = i < parts ? DS_LD(float2, pp + i, DS_BX_GNPART) : float2{0.f, 0.f};
    }
#else
    // range-checked buffer loads (pairs beyond `parts` read as zeros): with `i < parts ? load : 0` the compiler wraps every load in an
    // exec-masked region and sinks the float64 conversion AND its s_waitcnt vmcnt(0) into it — the prologue then pays one memory round trip
    // here before it has requested its tables, halo and weights
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp), (short)0, parts * 8, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2_t v = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (lane + 64 * k) * 8, 0, 0));
        g.v[k] = float2{v[0], v[1]};
    }
#endif
}
__device__ __forceinline__ void gn_partials_finish(const GnPartialLoads& g, const float* part, int parts, double count, float eps, int b, float& a, float& am) {
    const int lane = threadIdx.x & 63;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = g.v[k].x, y = g.v[k].y;
        asm volatile("" : "+v"(x), "+v"(y));        // the values are consumed HERE: the compiler otherwise hoists the conversions (and their
        s1 += (double)x;                             // s_waitcnt) up to the loads, in front of everything the caller issues in between
        s2 += (double)y;
    }
    const float* pp = part + (size_t)b * parts * 2;
    for (int i = lane + 256; i < parts; i += 64) {
        s1 += (double)DS_LD(float, pp + 2 * i, DS_BX_GNPART);
        s2 += (double)DS_LD(float, pp + 2 * i + 1, DS_BX_GNPART);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    a = (float)rstd;
    am = (float)(rstd * mean);
}

// GroupNorm(1,C) statistics straight from the producer's (sum, sumsq) partials: one wave, float64, no LDS.
// Returns rstd and rstd*mean (what ds_gn_finalize would have written to gn_ab).
// (bx: the operand the partials are checked against in the bounds build — a kernel that reduces TWO sets of partials names the second)
__device__ __forceinline__ void gn_from_partials(const float* part, int parts, double count, float eps, int b, float& a, float& am, int bx = DS_BX_GNPART) {
    const int lane = threadIdx.x & 63;
    double s1 = 0.0, s2 = 0.0;
    const float* pp = part + (size_t)b * parts * 2;
    (void)bx;
    for (int i = lane; i < parts; i += 64) {
        s1 += (double)DS_LD(float, pp + 2 * i, bx);
        s2 += (double)DS_LD(float, pp + 2 * i + 1, bx);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    a = (float)rstd;
    am = (float)(rstd * mean);
}
