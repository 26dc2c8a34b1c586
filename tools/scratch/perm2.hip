#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x, fh = lane >> 5, px = lane & 31;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 100.f * px + (8 * (r >> 2) + 4 * fh + (r & 3));   // value = pixel*100 + channel
    for (int q = 0; q < 2; ++q) {
        float v[8];
        for (int kk = 0; kk < 4; ++kk) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, acc[8 * q + kk]), __builtin_bit_cast(unsigned, acc[8 * q + 4 + kk]), false, false);
            v[kk] = __builtin_bit_cast(float, sw[0]);
            v[4 + kk] = __builtin_bit_cast(float, sw[1]);
        }
        for (int kk = 0; kk < 8; ++kk) out[px * 32 + 16 * q + 8 * fh + kk] = v[kk];
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 4096); k<<<1, 64>>>(d); float h[1024]; (void)hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < 32; ++p) for (int c = 0; c < 32; ++c) if (h[p * 32 + c] != 100.f * p + c) { if (bad < 8) printf("px %d ch %d got %g\n", p, c, h[p*32+c]); ++bad; }
    printf("bad=%d\n", bad);
}
