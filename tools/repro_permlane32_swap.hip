// Reproducer: __builtin_amdgcn_permlane32_swap unrolled over several registers emits ONE v_permlane32_swap_b32 (wrong results, bad=992)
// on ROCm 7.2 hipcc for gfx950; the inline-asm form (-DUSE_ASM) is correct (bad=0).  See conv_epilogue.hpp: permlane32_swap().
//   hipcc --offload-arch=gfx950 -O3 tools/repro_permlane32_swap.hip -o /tmp/a && /tmp/a ; same with -DUSE_ASM
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void swap32(float& a, float& b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__global__ void k(float* out, const float* in) {
    const int lane = threadIdx.x, fh = lane >> 5, px = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = in[lane * 16 + r];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#ifdef USE_ASM
            float a = acc[8 * q + kk], b = acc[8 * q + 4 + kk];
            swap32(a, b);
            v[kk] = a; v[4 + kk] = b;
#else
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, acc[8 * q + kk]), __builtin_bit_cast(unsigned, acc[8 * q + 4 + kk]), false, false);
            v[kk] = __builtin_bit_cast(float, sw[0]);
            v[4 + kk] = __builtin_bit_cast(float, sw[1]);
#endif
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) out[px * 32 + 16 * q + 8 * fh + kk] = v[kk];
    }
}
int main() {
    float *d, *di; (void)hipMalloc(&d, 4096); (void)hipMalloc(&di, 4096);
    float hi[1024];
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) hi[lane * 16 + r] = 100.f * (lane & 31) + (8 * (r >> 2) + 4 * (lane >> 5) + (r & 3));
    (void)hipMemcpy(di, hi, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, di); float h[1024]; (void)hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 32; ++p) for (int c = 0; c < 32; ++c) if (h[p * 32 + c] != 100.f * p + c) { if (bad < 4) printf("px %d ch %d got %g\n", p, c, h[p*32+c]); ++bad; }
    printf("bad=%d\n", bad);
}
