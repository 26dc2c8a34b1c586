// 3x3 stride-1 pad-1 convolution of an fp32 NHWC tensor to at most FOUR output channels (the U-Net's final_conv 96 -> 4,
// diffusion.py:103-105, in the fp32 and split-precision tiers; gfx950).
//
// On the generic fp32 implicit-GEMM tile this layer padded its 4 outputs to a 32-wide N tile of fp32 MFMAs and ran 1.37 ms per step at U-Net
// batch 128 (0.6 TB/s on a 0.8 GB input).  Four outputs are too few for a matrix tile and the layer is 3456 fused multiply-adds per pixel,
// so it runs on the vector ALUs: block = 16 x 16 pixels, thread = pixel; the 18 x 18 halo of one 32-channel chunk is staged in LDS (144-byte
// pixel pitch: conflict-free 16-byte reads with a thread per pixel), a thread reads its nine taps x 32 channels from LDS, and the chunk's
// 9 x 32 x 4 weights — identical for every lane — come through the constant address space as scalar loads, so every v_fma takes its weight
// from an SGPR (no LDS or vector-memory traffic for weights).  fp32 products and sums throughout: the tier's parity is unchanged.
#include "common.hpp"

namespace {

constexpr int N4_T = 16, N4_HP = N4_T + 2, N4_NPX = N4_HP * N4_HP, N4_PITCH = 144, N4_CH = 32;    // tile edge, halo edge, halo pixels, bytes per halo pixel
constexpr int N4_LDS = N4_NPX * N4_PITCH;                                                         // 46656
constexpr int N4_IT = (N4_NPX * 8 + 255) / 256;                                                   // 16-byte pieces of a chunk halo per thread: 11

typedef const float __attribute__((address_space(4))) * n4_cptr;

__global__ __launch_bounds__(256) void conv3x3_f32_n4_kernel(const float* x, int B, int H, int W, int C, const float* wpk, const float* bias, float* out,
                                                             int tiles_w, int tiles_hw) {
    __shared__ __attribute__((aligned(16))) char sm[N4_LDS];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h0 = th * N4_T, w0 = tw * N4_T;
    const int py = tid >> 4, px = tid & 15;
    const float* const xs = x + (size_t)b * H * W * C;
    (void)tiles_hw; (void)B;
    // halo pieces of this thread: piece id = tid + 256 it = (halo pixel, 16-byte quarter of its 128 channel bytes)
    int poff[N4_IT], plds[N4_IT];
#pragma unroll
    for (int it = 0; it < N4_IT; ++it) {
        const int id = tid + 256 * it, hp = id >> 3, q = id & 7;
        const int hr = hp / N4_HP, hc = hp - hr * N4_HP;
        const int hi = h0 + hr - 1, wi = w0 + hc - 1;
        const bool in = hp < N4_NPX && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        poff[it] = in ? (hi * W + wi) * C + 4 * q : -1;
        plds[it] = hp < N4_NPX ? hp * N4_PITCH + 16 * q : -1;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long wa = (unsigned long long)wpk;
    const n4_cptr wc = (n4_cptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wa >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)wa));
    const int nch = C / N4_CH;
    for (int cc = 0; cc < nch; ++cc) {
        // every piece of the chunk is requested before the first is written (pieces outside the image are zeros: the convolution's padding)
        f32x4 v[N4_IT];
#pragma unroll
        for (int it = 0; it < N4_IT; ++it) {
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (poff[it] >= 0) v[it] = DS_LD(f32x4, reinterpret_cast<const f32x4*>(xs + poff[it] + cc * N4_CH), DS_BX_SRC0);
        }
        if (cc) __syncthreads();                               // every thread is done with the previous chunk's image
#pragma unroll
        for (int it = 0; it < N4_IT; ++it)
            if (plds[it] >= 0) *reinterpret_cast<f32x4*>(sm + plds[it]) = v[it];
        __syncthreads();
        const n4_cptr wk = wc + (size_t)cc * 9 * N4_CH * 4;     // [tap][channel][4 outputs]
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* const hp = sm + ((py + t / 3) * N4_HP + px + t % 3) * N4_PITCH;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(hp + 16 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const n4_cptr w4 = wk + ((t * N4_CH + 4 * q + e) * 4);
                    acc[0] = fmaf(xv[e], w4[0], acc[0]);
                    acc[1] = fmaf(xv[e], w4[1], acc[1]);
                    acc[2] = fmaf(xv[e], w4[2], acc[2]);
                    acc[3] = fmaf(xv[e], w4[3], acc[3]);
                }
            }
        }
    }
    const int ho = h0 + py, wo = w0 + px;
    if (ho < H && wo < W) {
        const f32x4 r = f32x4{acc[0] + bias[0], acc[1] + bias[1], acc[2] + bias[2], acc[3] + bias[3]};
        DS_ST(f32x4, reinterpret_cast<f32x4*>(out + ((size_t)b * H * W + (size_t)ho * W + wo) * 4), DS_BX_OUT, r);
    }
}

// w [Cout <= 4][C][3][3] fp32 (Conv2d layout), bias [Cout] or NULL -> wpk [C / 32][9 taps][32 channels][4 outputs] and bias4 [4] (missing outputs: zeros)
__global__ void pack_conv3x3_f32_n4_kernel(const float* w, const float* bias, int Cout, int C, float* wpk, float* bias4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4) bias4[i] = (bias && i < Cout) ? bias[i] : 0.f;
    if (i >= C * 9 * 4) return;
    const int o = i & 3, c = (i >> 2) % N4_CH, t = (i / (4 * N4_CH)) % 9, cc = i / (4 * N4_CH * 9);
    const int ci = cc * N4_CH + c;
    wpk[i] = o < Cout ? w[((size_t)o * C + ci) * 9 + t] : 0.f;
}

}  // namespace

extern "C" size_t ds_conv3x3_f32_n4_weight_floats(int C) { return (size_t)C * 9 * 4 + 4; }

// dst: ds_conv3x3_f32_n4_weight_floats(C) floats = the packed weights followed by the four (zero-padded) biases
extern "C" int ds_pack_conv3x3_f32_n4(const float* w, const float* bias, int Cout, int C, float* dst, void* stream) {
    DS_REQUIRE(w && dst && Cout > 0 && Cout <= 4 && C > 0 && C % N4_CH == 0, "pack_conv3x3_f32_n4: Cout = %d must be at most 4, C = %d a multiple of %d", Cout, C, N4_CH);
    const int n = C * 9 * 4;
    hipLaunchKernelGGL(pack_conv3x3_f32_n4_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, bias, Cout, C, dst, dst + n);
    DS_CHECK_LAUNCH("pack_conv3x3_f32_n4");
    return DS_OK;
}

// out [B][H][W][4] fp32 = conv3x3(x [B][H][W][C] fp32, stride 1, zero padding 1) + bias, outputs beyond Cout are zero; wpk from ds_pack_conv3x3_f32_n4
extern "C" int ds_conv3x3_f32_n4(const float* x, int B, int H, int W, int C, const float* wpk, float* out, void* stream) {
    DS_REQUIRE(x && wpk && out && B > 0 && H > 0 && W > 0, "conv3x3_f32_n4: bad args");
    DS_REQUIRE(C > 0 && C % N4_CH == 0, "conv3x3_f32_n4: C = %d must be a multiple of %d", C, N4_CH);
    DS_REQUIRE((long long)H * W * C < (1ll << 31), "conv3x3_f32_n4: one input sample must have fewer than 2^31 elements");
    if (!ds_aligned16(x) || !ds_aligned16(wpk) || !ds_aligned16(out)) DS_FAIL(DS_EALIGN, "conv3x3_f32_n4: x / wpk / out must be 16-byte aligned");
    const int tiles_w = (W + N4_T - 1) / N4_T, tiles_h = (H + N4_T - 1) / N4_T;
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV3X3_F32_N4);
        h.set(DS_BX_SRC0, x, (long long)B * H * W * C * 4);
        h.set(DS_BX_OUT, out, (long long)B * H * W * 16);
        h.publish(reinterpret_cast<hipStream_t>(stream));
    }
#endif
    hipLaunchKernelGGL(conv3x3_f32_n4_kernel, dim3(tiles_w * tiles_h, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, B, H, W, C, wpk,
                       wpk + (size_t)C * 9 * 4, out, tiles_w, tiles_w * tiles_h);
    DS_CHECK_LAUNCH("conv3x3_f32_n4");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv3x3_f32_n4(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
