// 3x3 stride-1 pad-1 convolution, bf16 MFMA, direct from an LDS halo tile (gfx950).
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers the im2col A tile for each of the 9
// taps: 9x the global->LDS traffic and 9x the address arithmetic for the same input pixels.  Here a
// block owns a TH x TW patch of output pixels (TH*TW = 256) of one sample; per 32-channel chunk it
// stages the (TH+2) x (TW+2) input halo ONCE in LDS (64 B per pixel, XOR-swizzled by pixel index)
// and the 9 taps read their A fragments straight from it at shifted pixel offsets — no im2col tile
// is ever materialised.  Only the BN x 32 weight tile is streamed per (chunk, tap), double-buffered.
//   K order: channel-chunk major, tap minor; weights stay in the generic packed layout
//            [tap*NCC + cc][cout_pad][32], so one packing serves both kernels.
//   8 waves (512 threads): WM x WN waves of 32x32x16 bf16 MFMA tiles; block tile 256 x BN.
//   global->LDS bytes per MFMA flop: ~2.6x lower than the im2col kernel at BN = 192.
// Epilogue identical to conv_igemm.hip (GroupNorm fold, activation, residual, stats partials).
#include "common.hpp"
#include "conv_epilogue.hpp"

namespace {

constexpr int HALO_BYTES = 25600;  // >= 6*66*64 (TW=64), 10*34*64, 18*18*64, 34*10*64

__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int BN, int WM, int WN>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const ds_conv_params p, int twl) {
    constexpr int TM = 256 / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int B_BYTES = BN * 64;
    constexpr int B_IT = (BN * 4 + 511) / 512;
    static_assert(WM * WN == 8 && TM % 32 == 0 && TN % 32 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsH = smem;                   // [2][HALO_BYTES]
    char* const ldsB = smem + 2 * HALO_BYTES;  // [2][BN][32] bf16
    float* const red = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fh = lane >> 5;
    const int TW = 1 << twl, TH = 256 >> twl, HC = TW + 2, npx = (TH + 2) * HC;
    const int tiles_w = (p.W + TW - 1) >> twl;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const int b = blockIdx.z, n0 = blockIdx.y * BN;
    const int Cin = p.C0, NCC = Cin >> 5;

    const bf16* src = reinterpret_cast<const bf16*>(p.src0) + (size_t)b * p.H * p.W * Cin;
    const bf16* wbase = reinterpret_cast<const bf16*>(p.wpk) + (size_t)n0 * 32 + tid * 8;
    const size_t wstride = (size_t)p.cout_pad * 32;  // elements per packed K chunk

    // ---- halo loader: slot = tid + it*512 -> (pixel, 16-B chunk); fixed per thread for the whole kernel
    int hoff[4], hlds[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = tid + it * 512, px = slot >> 2, ch = slot & 3;
        hoff[it] = -1;
        hlds[it] = -1;
        if (px < npx) {
            const int hr = px / HC, hc = px - hr * HC;
            const int hi = h0 + hr - 1, wi = w0 + hc - 1;
            hlds[it] = swz64(px, ch);
            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) hoff[it] = (hi * p.W + wi) * Cin + ch * 8;
        }
    }
    uint4 rh[4], rb[B_IT];
    auto load_halo = [&](int cc) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            rh[it] = hoff[it] >= 0 ? *reinterpret_cast<const uint4*>(src + hoff[it] + cc * 32) : make_uint4(0, 0, 0, 0);
    };
    auto store_halo = [&](int buf) {
        char* h = ldsH + buf * HALO_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (hlds[it] >= 0) *reinterpret_cast<uint4*>(h + hlds[it]) = rh[it];
    };
    auto load_b = [&](int cc, int tap) {
        const bf16* w = wbase + (size_t)(tap * NCC + cc) * wstride;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const bool ok = (it + 1) * 512 <= BN * 4 || tid + it * 512 < BN * 4;
            rb[it] = ok ? *reinterpret_cast<const uint4*>(w + it * 512 * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_b = [&](int buf) {
        char* bb = ldsB + buf * B_BYTES;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int slot = tid + it * 512;
            if ((it + 1) * 512 <= BN * 4 || slot < BN * 4) *reinterpret_cast<uint4*>(bb + swz64(slot >> 2, slot & 3)) = rb[it];
        }
    };

    // ---- per-lane fragment bases
    int p0[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * TM + i * 32 + frow;
        p0[i] = (ml >> twl) * HC + (ml & (TW - 1));
    }
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int hbuf, int bbuf, int kh, int kw) {
        const char* h = ldsH + hbuf * HALO_BYTES;
        const char* bb = ldsB + bbuf * B_BYTES;
        const int shift = kh * HC + kw;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(h + swz64(p0[i] + shift, 2 * s + fh));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(bb + swz64(wn * TN + j * 32 + frow, 2 * s + fh));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: channel chunks x 9 taps; B double-buffered per step, halo double-buffered per chunk
    load_halo(0);
    load_b(0, 0);
    store_halo(0);
    store_b(0);
    __syncthreads();
    for (int cc = 0; cc < NCC; ++cc) {
        const bool more_cc = cc + 1 < NCC;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = cc * 9 + tap;
            const bool more = tap < 8 || more_cc;
            if (more) load_b(tap < 8 ? cc : cc + 1, tap < 8 ? tap + 1 : 0);
            if (tap == 6 && more_cc) load_halo(cc + 1);
            compute(cc & 1, s & 1, tap / 3, tap % 3);
            if (more) store_b((s + 1) & 1);
            if (tap == 8 && more_cc) store_halo((cc + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue (conv_epilogue.hpp)
    auto coord = [&](int ml) {
        ConvCoord c;
        c.ho = h0 + (ml >> twl);
        c.wo = w0 + (ml & (TW - 1));
        c.ok = c.ho < p.H && c.wo < p.W;
        c.pix = c.ho * p.W + c.wo;
        return c;
    };
    float s1 = 0.f, s2 = 0.f;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * (TN + 4));
    conv_epilogue<bf16, FM, FN>(p, acc, b, n0 + wn * TN, wm * TM, p.H * p.W, stage, coord, s1, s2);
    __syncthreads();
    if (p.stats_part) {
        const int parts = gridDim.x * gridDim.y;
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2);
    }
}

int halo_twl(int W) {
    // patch width: smallest power of two >= min(W, 64), at least 8
    int twl = 3;
    while ((1 << twl) < W && twl < 6) ++twl;
    return twl;
}

template <int BN, int WM, int WN>
int launch_halo(const ds_conv_params& p, hipStream_t st) {
    constexpr size_t lds_main = 2 * (size_t)HALO_BYTES + 2 * (size_t)BN * 64;
    constexpr size_t lds_epi = 8 * 32 * (size_t)(BN / WN + 4) * sizeof(float);
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    auto kern = conv3x3_halo_kernel<BN, WM, WN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) DS_FAIL(DS_ELAUNCH, "conv3x3_halo: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
        attr_done = true;
    }
    const int twl = halo_twl(p.W), TW = 1 << twl, TH = 256 >> twl;
    dim3 grid(((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW), p.cout_pad / BN, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p, twl);
    DS_CHECK_LAUNCH("conv3x3_halo");
    return DS_OK;
}

}  // namespace

// called from ds_conv_igemm for tile ids DS_CONV_TILE_HALO_*
int ds_conv3x3_halo_parts(const ds_conv_params* p) {
    const int twl = halo_twl(p->W), TW = 1 << twl, TH = 256 >> twl;
    const int bn = p->tile == DS_CONV_TILE_HALO_256x192 ? 192 : 96;
    return ((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW) * (p->cout_pad / bn);
}

int ds_conv3x3_halo_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv3x3_halo: bf16 only");
    DS_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad_h == 1 && p->pad_w == 1 && !p->transposed,
               "conv3x3_halo: 3x3 stride 1 pad 1 only");
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0, "conv3x3_halo: single source, Cin multiple of 32 (got %d+%d)", p->C0, p->C1);
    DS_REQUIRE(p->Ho == p->H && p->Wo == p->W && !p->out_nchw_f32, "conv3x3_halo: same-size NHWC output only");
    if (p->tile == DS_CONV_TILE_HALO_256x192) return launch_halo<192, 4, 2>(*p, st);
    return launch_halo<96, 8, 1>(*p, st);
}
