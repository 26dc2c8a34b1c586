// Micro-benchmark (diagnostic, not part of the product): what does a VALU stream cost beside an MFMA stream on the SAME SIMD?
// One 512-thread block per CU: waves 0-3 and 4-7 pair up on the four SIMDs.  Role A (waves 0-3) issues v_mfma_f32_16x16x32_bf16
// back to back; role B (waves 4-7) issues one kind of VALU instruction.  Each role is timed alone and beside the other.
//   hipcc --offload-arch=gfx950 -O3 -o coissue tools/ubench/coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

template <int KIND>
__device__ __forceinline__ void valu_body(float (&r)[16], float k) {
    // 16 independent chains, one instruction each per call
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 1) {
            if (i % 2 == 0) {
                f32x2 v = {r[i], r[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(f32x2{k, k}));
                r[i] = v[0]; r[i + 1] = v[1];
            }
        }
        if constexpr (KIND == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        if constexpr (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 6) {
            if (i % 2 == 0) {
                f32x2 v = {r[i], r[i + 1]};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(f32x2{k, k}));
                r[i] = v[0]; r[i + 1] = v[1];
            }
        }
        if constexpr (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(f32x2*)&r[i & ~1]) : "v"(f32x2{k, k}));
        if constexpr (KIND == 8) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 9) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 10) {
            if (i % 2 == 0) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        }
        if constexpr (KIND == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 14) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
        if constexpr (KIND == 16) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 17) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[i]) : "v"(k));
        if constexpr (KIND == 18) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(r[i]) : "v"(k));
    }
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool roleA = wave < 4;
    const bool active = roleA ? (mode & 1) : (mode & 2);
    long t0 = 0, t1 = 0;
    if (roleA) {
        f32x4 acc[8];
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane ^ i)); }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (active)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.456f) sink[threadIdx.x] = s;
    } else {
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = 0.5f + 0.001f * (lane + i);
        const float kk = 0.999f;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (active)
            for (int it = 0; it < iters; ++it) {
                valu_body<KIND>(r, kk);
                valu_body<KIND>(r, kk);
            }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += r[i];
        if (s == 123.456f) sink[threadIdx.x] = s;
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_call) {
    const int iters = 2000, nblk = 256;
    long* d; float* sink;
    hipMalloc(&d, nblk * 8 * sizeof(long)); hipMalloc(&sink, 4096);
    std::vector<long> h(nblk * 8);
    double res[4][2] = {};
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(512), 0, 0, mode, iters, d, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, h.size() * sizeof(long), hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < nblk; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w];
        res[mode][0] = a / (nblk * 4) / (iters * 32.0);
        res[mode][1] = b / (nblk * 4) / (iters * 2.0 * per_call);
    }
    printf("%-18s  mfma alone %.1f cyc/mfma | valu alone %.1f cyc/op | together: mfma %.1f cyc/mfma, valu %.1f cyc/op (until each finished its own count)\n",
           name, res[1][0], res[2][1], res[3][0], res[3][1]);
    hipFree(d); hipFree(sink);
}

int main() {
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_pk_fma_f16", 16);
    run<3>("v_exp_f32", 16);
    run<4>("v_cvt_pk_bf16_f32", 16);
    run<5>("v_max_f32", 16);
    run<6>("v_pk_add_f32", 8);
    run<8>("v_dot2c_f32_bf16", 16);
    run<9>("v_max3_f32", 16);
    run<10>("v_permlane32_swap", 8);
    run<11>("v_add_f32", 16);
    run<12>("v_mov_b32", 16);
    run<13>("v_cndmask_b32", 16);
    run<14>("v_fmac_f32", 16);
    run<15>("v_rcp_f32", 16);
    run<16>("v_cvt_pkrtz_f16_f32", 16);
    run<17>("v_pk_mul_f16", 16);
    run<18>("v_dot2c_f32_f16", 16);
    return 0;
}
