// Implicit-GEMM convolution on MFMA for channels-last activations (gfx950).
//
//   C[m][n] = sum_k A[m][k] * Wp[n][k]      m = output pixel of ONE sample (per-sample M tiling),
//                                            n = output channel, k = tap*Cin + c
// A is gathered on the fly from one or two NHWC sources (zero-copy skip concat, zero padding by
// predication), weights come pre-packed as [kchunk][cout_pad][32] so a block's B tile is one
// contiguous run.  K is walked in chunks of 32; per chunk the block stages BM x 32 of A and
// BN x 32 of B through registers into double-buffered, XOR-swizzled LDS (one barrier per chunk),
// and each of the 4 waves owns a (BM/WM) x (BN/WN) sub-tile of 32x32 MFMA accumulators:
//   bf16: v_mfma_f32_32x32x16_bf16   (2 per chunk and accumulator)
//   fp32: v_mfma_f32_32x32x2_f32     (16 per chunk and accumulator; exact fp32 fma chain)
// Epilogue (registers only): GroupNorm(1,C)-of-the-input fold  v = a_b*acc + (t1[cls][n] - a_b*m_b*t2[cls][n]),
// or bias; activation; residual; store NHWC (or fp32 NCHW); per-block (sum, sumsq) partial for the next norm.
#include <type_traits>

#include "common.hpp"
#include "conv_epilogue.hpp"

namespace {

inline bool is_halo_tile(int tile) { return tile == DS_CONV_TILE_HALO3_256x96; }

template <typename T> struct Lds;
template <> struct Lds<float> {
    static constexpr int CPR = 8;   // 16-B chunks per 32-element row
    static constexpr int RPB = 2;   // rows per 256-B LDS bank row
    static constexpr int RB = 128;  // row bytes
};
template <> struct Lds<bf16> {
    static constexpr int CPR = 4;
    static constexpr int RPB = 4;
    static constexpr int RB = 64;
};

template <typename T> __device__ __forceinline__ int swz(int row, int chunk) {
    // byte offset of 16-B chunk `chunk` of tile row `row`; conflict-free for the ds_read_b128 lane groups
    return row * Lds<T>::RB + ((chunk ^ ((row / Lds<T>::RPB) & (Lds<T>::CPR - 1))) << 4);
}

template <typename T> __device__ __forceinline__ void store_out(T* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16>(bf16* p, float v) { *p = (bf16)v; }

// (the fp32 256 x 96 tile needs more than 256 registers with its two-stage ring: one block of it per CU)
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, (sizeof(T) == 4 && BM == 256) ? 1 : 2) void conv_igemm_kernel(const ds_conv_params p) {
    using L = Lds<T>;
    constexpr int EPC = ElemTr<T>::EPC;
    constexpr int CPR = L::CPR;
    constexpr int RSTEP = 256 / CPR;
    constexpr int A_IT = (BM + RSTEP - 1) / RSTEP;   // last iteration may be partial (guarded)
    constexpr int B_IT = (BN + RSTEP - 1) / RSTEP;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM % 32 == 0 && TN % 32 == 0, "tile shape");
    constexpr int A_BYTES = BM * L::RB, B_BYTES = BN * L::RB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;                 // [2][BM][32]
    char* const ldsB = smem + 2 * A_BYTES;   // [2][BN][32]
    float* const red = reinterpret_cast<float*>(smem);  // reused after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nphase = p.transposed ? 4 : 1;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int zb = blockIdx.z / ksplit, kz = blockIdx.z - zb * ksplit;   // split-K: this block owns K steps [q_lo, q_hi)
    const int b = zb / nphase, phase = zb % nphase;
    const int pa = phase >> 1, pb = phase & 1;
    const int HoWo = p.Ho * p.Wo;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int Cin = p.C0 + p.C1;
    const int pad_h = p.transposed ? 1 - pa : p.pad_h;
    const int pad_w = p.transposed ? 1 - pb : p.pad_w;
    const int ntap = p.KH * p.KW;
    const int nq = (ntap * Cin + 31) / 32;

    float gn_a = 1.f, gn_am = 0.f;

    const T* src0 = reinterpret_cast<const T*>(p.src0) + (size_t)b * p.H * p.W * p.C0;
    const T* src1 = p.C1 ? reinterpret_cast<const T*>(p.src1) + (size_t)b * p.H1 * p.W1 * p.C1 : nullptr;
    // (threads whose row lies past a narrow BN tile fetch row 0 instead: their loads are unconditional but never stored)
    const T* wq = reinterpret_cast<const T*>(p.wpk) + ((size_t)phase * nq * p.cout_pad + n0) * 32 +
                  (tid * EPC / 32 < BN ? tid * EPC : (tid * EPC) % 32);

    // ---- loader state: this thread owns 16-B chunk `ch` of rows r0 + i*RSTEP -------------------------
    const int ch = tid % CPR, r0 = tid / CPR;
    int hb[A_IT], wb[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + i * RSTEP;
        if (m < HoWo && r0 + i * RSTEP < BM) {
            const int ho = m / p.Wo, wo = m - ho * p.Wo;
            hb[i] = ho * p.stride - pad_h;
            wb[i] = wo * p.stride - pad_w;
        } else {
            hb[i] = -(1 << 28);
            wb[i] = -(1 << 28);
        }
    }
    const int nqs = (nq + ksplit - 1) / ksplit, q_lo = kz * nqs, q_hi = min(nq, q_lo + nqs);
    int kc = (q_lo * 32 + ch * EPC) % Cin, tap = (q_lo * 32 + ch * EPC) / Cin;
    int tap_h = tap / p.KW;

    // K tiles travel through a PF-slot register ring, PF-1 steps ahead of the MFMAs that consume them: with a single
    // stage the ~2 us of a global load under load were exposed on every 32-deep K step (the 1x1 / 4x4 / transposed
    // layers ran at 80-300 TFLOP/s).  Every load is unconditional (clamped address + select): a load under a branch
    // makes hipcc drain the whole ring with s_waitcnt vmcnt(0) at the merge point.
    constexpr int PF = sizeof(T) == 2 ? 3 : 2;
    u32x4 ra[PF][A_IT], rb[PF][B_IT];
    int qload = q_lo;
    int boff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) boff[i] = ((i + 1) * RSTEP <= BN || r0 + i * RSTEP < BN) ? i * 256 * EPC : 0;
    auto load_tiles = [&](auto slotc) {
        constexpr int sl = decltype(slotc)::value;
        const int kh = tap_h, kw = tap - tap_h * p.KW;   // (kh, kw) tracked incrementally: no integer division in the K loop
        const bool tap_ok = tap < ntap && qload < q_hi;  // false for the dummy tiles past the end of this block's K range
        const bool first = kc < p.C0;
        const T* base = first ? src0 : src1;
        const int Cs = first ? p.C0 : p.C1, cc = first ? kc : kc - p.C0;
        const int Hs = first ? p.H : p.H1, Ws = first ? p.W : p.W1;
        const int dh = first ? kh : kh - p.off_h1, dw = first ? kw : kw - p.off_w1;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int hi = hb[i] + dh, wi = wb[i] + dw;
            const bool ok = tap_ok && (unsigned)hi < (unsigned)Hs && (unsigned)wi < (unsigned)Ws;
            const u32x4 v = DS_LD(u32x4, ok ? base + ((size_t)(hi * Ws + wi) * Cs + cc) : src0, (ok && !first) ? DS_BX_SRC1 : DS_BX_SRC0);
            ra[sl][i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
        // dummy tiles (past this block's K range) re-read a chunk that certainly exists: q_lo itself lies past the packed
        // weights when a split-K slice is empty (q_lo >= nq), hence the second clamp
        const T* wsrc = wq + (size_t)(qload < q_hi ? qload : (q_lo < nq ? q_lo : nq - 1)) * p.cout_pad * 32;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[sl][i] = DS_LD(u32x4, wsrc + boff[i], DS_BX_W);
        ++qload;
        kc += 32;
        while (kc >= Cin) {
            kc -= Cin;
            ++tap;
            if (tap - tap_h * p.KW >= p.KW) ++tap_h;
        }
    };
    auto store_tiles = [&](auto slotc, int buf) {
        constexpr int sl = decltype(slotc)::value;
        char* a = ldsA + buf * A_BYTES;
        char* bb = ldsB + buf * B_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if ((i + 1) * RSTEP <= BM || r0 + i * RSTEP < BM) *reinterpret_cast<u32x4*>(a + swz<T>(r0 + i * RSTEP, ch)) = ra[sl][i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((i + 1) * RSTEP <= BN || r0 + i * RSTEP < BN) *reinterpret_cast<u32x4*>(bb + swz<T>(r0 + i * RSTEP, ch)) = rb[sl][i];
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
        const char* a = ldsA + buf * A_BYTES;
        const char* bb = ldsB + buf * B_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[FM], bf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(a + swz<T>(wm * TM + i * 32 + frow, 2 * s + fh));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(bb + swz<T>(wn * TN + j * 32 + frow, 2 * s + fh));
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // 32x32x2 f32: lane half h supplies k = 8u + 4h + e for MFMA e of group u (A and B agree,
            // every k of the chunk is covered exactly once), so one ds_read_b128 feeds four MFMAs.
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 af[FM], bf[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(a + swz<T>(wm * TM + i * 32 + frow, 2 * u + fh));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(bb + swz<T>(wn * TN + j * 32 + frow, 2 * u + fh));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop: PF-slot register ring -> double-buffered LDS, one barrier per 32-deep K step.  K is padded to a
    // multiple of PF steps with all-zero dummy tiles so that the unrolled ring needs no branch.
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    load_tiles(R0{});
    load_tiles(R1{});
    if constexpr (PF == 3) load_tiles(R2{});
    // GroupNorm statistics of the input from the producer's partials, overlapped with the first tile loads
    if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, gn_a, gn_am);
    store_tiles(R0{}, 0);
    __syncthreads();
    const int nqp = (q_hi - q_lo + PF - 1) / PF * PF;
    auto step = [&](auto sc, int q) {
        constexpr int sl = decltype(sc)::value;
        load_tiles(sc);                                                   // slot sl went to LDS in the previous step
        compute(q & 1);
        store_tiles(std::integral_constant<int, (sl + 1) % PF>{}, (q + 1) & 1);
        __syncthreads();
    };
    for (int q0 = 0; q0 < nqp; q0 += PF) {
        step(R0{}, q0);
        step(R1{}, q0 + 1);
        if constexpr (PF == 3) step(R2{}, q0 + 2);
    }

    // ---- epilogue (conv_epilogue.hpp): wave-private LDS transpose, 16-byte row-major stores
    const int outW = p.transposed ? 2 * p.Wo : p.Wo;
    const int outHW = p.transposed ? 4 * HoWo : HoWo;
    const bool need_hw = p.ncls == 9 || p.transposed;   // plain GEMM-like launches never divide
    auto coord = [&](int ml) {
        ConvCoord c;
        const int m = m0 + ml;
        c.ok = m < HoWo;
        c.ho = c.wo = 0;
        c.pix = m;
        if (need_hw) {
            c.ho = m / p.Wo;
            c.wo = m - c.ho * p.Wo;
            if (p.transposed) c.pix = (2 * c.ho + pa) * outW + 2 * c.wo + pb;
        }
        return c;
    };
    float s1 = 0.f, s2 = 0.f;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * (TN + 4));
    if (ksplit > 1) {
        // raw fp32 partial sums of this K slice -> slab[kz][b]; bias / fold / activation / residual / statistics
        // happen in ds_conv_splitk_reduce
        ds_conv_params q = p;
        q.out = p.slab;
        q.out_C = (p.Cout + 7) / 8 * 8;
        q.out_c0 = 0;
        q.bias = nullptr; q.gn_ab = nullptr; q.gn_part = nullptr; q.res = nullptr;
        conv_epilogue_body<float, FM, FN, DS_ACT_NONE, false>(q, acc, kz * p.B + b, n0 + wn * TN, wm * TM, outHW, stage, coord, s1, s2);
        return;
    }
    if constexpr (!(DS_ABLATE & 64)) conv_epilogue<T, FM, FN>(p, acc, b, n0 + wn * TN, wm * TM, outHW, stage, coord, s1, s2, gn_a, gn_am);
    __syncthreads();   // stage regions overlap `red`
    if (p.stats_part) {
        const int parts = gridDim.x * gridDim.y * nphase;
        const int slot = (phase * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * parts + slot) * 2);
    }
}

}  // namespace
#if DS_BOUNDS
// byte extents of every operand of a convolution launch, from the parameter struct alone (shared with conv3x3_halo3.hip and conv_splitk.hip)
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out) {
    DsBxHost h(kernel);
    const long long es = p.dtype == DS_BF16 ? 2 : 4;
    const long long oH = p.transposed ? 2 * p.Ho : p.Ho, oW = p.transposed ? 2 * p.Wo : p.Wo;
    const int Cin = p.C0 + p.C1;
    h.set(DS_BX_SRC0, p.src0, (long long)p.B * p.H * p.W * p.C0 * es);
    h.set(DS_BX_SRC1, p.C1 ? p.src1 : nullptr, (long long)p.B * p.H1 * p.W1 * p.C1 * es);
    h.set(DS_BX_W, p.wpk, (long long)ds_pack_conv_elems(Cin, p.KH, p.KW, p.cout_pad, p.transposed) * es);
    if (p.ksplit > 1) h.set(DS_BX_OUT, p.slab, (long long)p.ksplit * p.B * oH * oW * ((p.Cout + 7) / 8 * 8) * 4);
    else h.set(DS_BX_OUT, p.out, (long long)p.B * oH * oW * p.out_C * es);
    h.set(DS_BX_RES, p.res, (long long)p.B * oH * oW * p.out_C * es);
    h.set(DS_BX_BIAS, p.bias, (long long)p.Cout * 4);
    h.set(DS_BX_T1, p.fold_t1, (long long)p.ncls * p.Cout * 4);
    h.set(DS_BX_T2, p.fold_t2, (long long)p.ncls * p.Cout * 4);
    h.set(DS_BX_GNAB, p.gn_ab, (long long)p.B * 2 * 4);
    h.set(DS_BX_GNPART, p.gn_part, (long long)p.B * p.gn_parts * 2 * 4);
    h.set(DS_BX_STATS, p.stats_part, (long long)p.B * stats_parts * 2 * 4);
    *out = h.t;
}
static void ds_conv_publish_bounds(const ds_conv_params& p, int kernel, int stats_parts, hipStream_t st) {
    DsBxHost h(kernel);
    ds_conv_bounds_table(p, kernel, stats_parts, &h.t);
    h.publish(st);
}
extern "C" int ds_bounds_fetch_conv_igemm(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
// negative control of the tool itself: one 16-byte load that starts 8 bytes before the end of a 64-byte extent
__global__ void ds_bounds_selftest_kernel(const float* buf, float* sink) {
    const f32x4 v = DS_LD(f32x4, buf + 14, DS_BX_AUX3);
    if (v[0] == 12345.f) *sink = v[1];
}
extern "C" int ds_bounds_selftest(const float* buf64, float* sink, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DsBxHost h(DS_K_CONV_IGEMM);
    h.set(DS_BX_AUX3, buf64, 64);
    h.publish(st);
    hipLaunchKernelGGL(ds_bounds_selftest_kernel, dim3(1), dim3(1), 0, st, buf64, sink);
    DS_CHECK_LAUNCH("bounds_selftest");
    return DS_OK;
}
#endif
namespace {

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const ds_conv_params& p, hipStream_t st) {
    constexpr size_t lds_main = 2 * (size_t)(BM + BN) * Lds<T>::RB;
    constexpr size_t lds_epi = 4 * 32 * (size_t)(BN / WN + 4) * sizeof(float);   // 4 wave-private transpose stages
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN>;
    DS_SET_MAX_LDS(kern, lds, "conv_igemm");
    const int HoWo = p.Ho * p.Wo;
    dim3 grid((HoWo + BM - 1) / BM, p.cout_pad / BN, p.B * (p.transposed ? 4 : 1) * (p.ksplit > 1 ? p.ksplit : 1));
#if DS_BOUNDS
    ds_conv_publish_bounds(p, DS_K_CONV_IGEMM, grid.x * grid.y * (p.transposed ? 4 : 1), st);
#endif
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    DS_CHECK_LAUNCH("conv_igemm");
    return DS_OK;
}

template <typename T> int launch_tile(const ds_conv_params& p, hipStream_t st) {
    switch (p.tile) {
        case DS_CONV_TILE_128x192: return launch_cfg<T, 128, 192, 2, 2>(p, st);
        case DS_CONV_TILE_256x96: return launch_cfg<T, 256, 96, 4, 1>(p, st);
        case DS_CONV_TILE_128x32: return launch_cfg<T, 128, 32, 4, 1>(p, st);
        case DS_CONV_TILE_64x96: return launch_cfg<T, 64, 192, 2, 2>(p, st);
    }
    DS_FAIL(DS_EINVAL, "conv_igemm: unknown tile %d", p.tile);
}

void tile_dims(int tile, int* bm, int* bn) {
    switch (tile) {
        case DS_CONV_TILE_QUAD_HALO3:
        case DS_CONV_TILE_HALO3_256x96: *bm = 256; *bn = 96; break;
        case DS_CONV_TILE_128x192: *bm = 128; *bn = 192; break;
        case DS_CONV_TILE_256x96: *bm = 256; *bn = 96; break;
        case DS_CONV_TILE_128x32: *bm = 128; *bn = 32; break;
        case DS_CONV_TILE_HALO3_N16: *bm = 256; *bn = 16; break;
        case DS_CONV_TILE_64x96: *bm = 64; *bn = 192; break;
        default: *bm = 0; *bn = 0;          // (ids 4 .. 10: the retired first / second generation halo kernels)
    }
}

int validate(const ds_conv_params* p) {
    DS_REQUIRE(p != nullptr, "conv_igemm: null params");
    int bm, bn;
    tile_dims(p->tile, &bm, &bn);
    DS_REQUIRE(bm > 0, "conv_igemm: unknown tile %d", p->tile);
    const int epc = p->dtype == DS_BF16 ? 8 : 4;
    DS_REQUIRE(p->dtype == DS_F32 || p->dtype == DS_BF16, "conv_igemm: dtype %d", p->dtype);
    DS_REQUIRE(p->B > 0 && p->Ho > 0 && p->Wo > 0 && p->H > 0 && p->W > 0, "conv_igemm: empty problem");
    DS_REQUIRE(p->C0 > 0 && p->C0 % epc == 0 && p->C1 >= 0 && p->C1 % epc == 0,
               "conv_igemm: channel counts (%d,%d) must be multiples of %d", p->C0, p->C1, epc);
    DS_REQUIRE(p->C1 == 0 || (p->src1 && p->H1 > 0 && p->W1 > 0), "conv_igemm: second source incomplete");
    DS_REQUIRE(p->cout_pad % bn == 0 && p->Cout <= p->cout_pad && p->Cout > 0,
               "conv_igemm: cout_pad %d must be a multiple of the tile's BN %d", p->cout_pad, bn);
    DS_REQUIRE(p->KH > 0 && p->KW > 0 && p->stride > 0, "conv_igemm: bad geometry");
    DS_REQUIRE(!p->transposed || (p->KH == 2 && p->KW == 2 && p->stride == 1), "conv_igemm: transposed needs KH=KW=2");
    DS_REQUIRE(p->ncls == 1 || p->ncls == 9, "conv_igemm: ncls must be 1 or 9");
    DS_REQUIRE(!(p->gn_ab || p->gn_part) || (p->fold_t1 && p->fold_t2), "conv_igemm: GN fold needs t1/t2 tables");
    DS_REQUIRE(!p->gn_part || (p->gn_parts > 0 && p->gn_count > 0), "conv_igemm: gn_part needs gn_parts and gn_count");
    DS_REQUIRE(p->ncls == 1 || (p->KH == 3 && p->KW == 3 && p->pad_h == 1 && p->pad_w == 1 && p->stride == 1 &&
                                p->Ho == p->H && p->Wo == p->W && p->H >= 2 && p->W >= 2),
               "conv_igemm: 9 border classes are defined for 3x3 pad 1 stride 1 only");
    DS_REQUIRE(!p->out_nchw_f32, "conv_igemm: NCHW output was removed from the kernel; convert with ds_nhwc_to_nchw");
    DS_REQUIRE(p->act == DS_ACT_NONE || p->act == DS_ACT_GELU, "conv_igemm: epilogue activation must be NONE or GELU (got %d)", p->act);
    DS_REQUIRE(p->out_C % epc == 0 && p->out_c0 % epc == 0 && p->out_C >= p->out_c0 + (p->Cout + epc - 1) / epc * epc,
               "conv_igemm: out_C=%d / out_c0=%d must be multiples of %d and hold Cout=%d rounded up", p->out_C, p->out_c0, epc, p->Cout);
    if (!ds_aligned16(p->src0) || !ds_aligned16(p->wpk) || (p->C1 && !ds_aligned16(p->src1)))
        DS_FAIL(DS_EALIGN, "conv_igemm: src/weight pointers must be 16-byte aligned");
    DS_REQUIRE(p->out != nullptr, "conv_igemm: null output");
    DS_REQUIRE(p->ksplit <= 1 || (p->dtype == DS_BF16 && p->slab && (p->ksplit == 2 || p->ksplit == 3 || p->ksplit == 4 || p->ksplit == 6 || p->ksplit == 8)),
               "conv_igemm: split-K needs bf16, a slab and ksplit in {2, 3, 4, 6, 8} (got %d)", p->ksplit);
    DS_REQUIRE(p->ksplit <= 1 || is_halo_tile(p->tile) ||
                   ((p->transposed ? 4 : p->KH * p->KW) * (p->C0 + p->C1) + 31) / 32 >= 2 * p->ksplit,
               "conv_igemm: ksplit=%d leaves fewer than two K steps per slice", p->ksplit);
    if (p->ksplit > 1 && !is_halo_tile(p->tile)) {
        const int nq = ((p->transposed ? 4 : p->KH * p->KW) * (p->C0 + p->C1) + 31) / 32, nqs = (nq + p->ksplit - 1) / p->ksplit;
        DS_REQUIRE((p->ksplit - 1) * nqs < nq, "conv_igemm: ksplit=%d over %d K steps leaves the last slice empty", p->ksplit, nq);
    }
    DS_REQUIRE((p->wk_order == 2) == (p->tile == DS_CONV_TILE_QUAD_HALO3), "conv_igemm: wk_order=%d does not match tile %d (quad tiles are for DS_CONV_TILE_QUAD_HALO3 only)",
               p->wk_order, p->tile);
    DS_REQUIRE((p->wk_order == 1) == (p->tile == DS_CONV_TILE_HALO3_256x96 || p->tile == DS_CONV_TILE_HALO3_N16),
               "conv_igemm: wk_order=%d does not match tile %d (chunk-major weights are for DS_CONV_TILE_HALO3_256x96 / _N16 only)", p->wk_order, p->tile);
    // fields only some tiles read: anywhere else they must be 0 (a caller who sets them would get a plain convolution, silently)
    DS_REQUIRE(p->flags == 0 || p->tile == DS_CONV_TILE_HALO3_256x96 || p->tile == DS_CONV_TILE_QUAD_HALO3,
               "conv_igemm: flags=%d (split precision) is implemented by DS_CONV_TILE_HALO3_256x96 / QUAD_HALO3 only, not tile %d", p->flags, p->tile);
    DS_REQUIRE(p->res_steps == 0 || p->tile == DS_CONV_TILE_HALO3_256x96,
               "conv_igemm: a fused res_conv (res_steps=%d) is implemented by DS_CONV_TILE_HALO3_256x96 only, not tile %d", p->res_steps, p->tile);
    DS_REQUIRE(p->res_steps != 0 || (!p->res_src0 && !p->res_src1 && !p->res_bias),
               "conv_igemm: res_src0 / res_src1 / res_bias given with res_steps = 0");
    return DS_OK;
}

}  // namespace

int ds_conv_splitk_parts(const ds_conv_params* p);                       // conv_splitk.hip
int ds_conv3x3_halo3_launch(const ds_conv_params* p, hipStream_t st);   // conv3x3_halo3.hip
int ds_conv3x3_halo3_parts(const ds_conv_params* p);
int ds_conv_quad_halo3_launch(const ds_conv_params* p, hipStream_t st);  // conv_quad_halo3.hip
int ds_conv_quad_halo3_parts(const ds_conv_params* p);
int ds_conv3x3_smalln_launch(const ds_conv_params* p, hipStream_t st);   // conv3x3_smalln.hip

extern "C" int ds_conv_tile_bn(int tile) {
    int bm, bn;
    tile_dims(tile, &bm, &bn);
    return bn;
}

extern "C" int ds_conv_stats_parts(const ds_conv_params* p) {
    int bm, bn;
    tile_dims(p->tile, &bm, &bn);
    if (!bm) return DS_EINVAL;
    if (p->ksplit > 1) return ds_conv_splitk_parts(p);
    if (p->tile == DS_CONV_TILE_QUAD_HALO3) return ds_conv_quad_halo3_parts(p);
    if (p->tile == DS_CONV_TILE_HALO3_256x96) return ds_conv3x3_halo3_parts(p);
    return ((p->Ho * p->Wo + bm - 1) / bm) * (p->cout_pad / bn) * (p->transposed ? 4 : 1);
}

extern "C" int ds_conv_igemm(const ds_conv_params* p, void* stream) {
    int rc = validate(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p->tile == DS_CONV_TILE_HALO3_256x96) return ds_conv3x3_halo3_launch(p, st);
    if (p->tile == DS_CONV_TILE_QUAD_HALO3) return ds_conv_quad_halo3_launch(p, st);
    if (p->tile == DS_CONV_TILE_HALO3_N16) return ds_conv3x3_smalln_launch(p, st);
    return p->dtype == DS_BF16 ? launch_tile<bf16>(*p, st) : launch_tile<float>(*p, st);
}

// ------------------------------------------------------------------------------------------------ packing
namespace {

template <typename T>
__global__ void pack_conv_kernel(const ds_pack_conv_params p, int nq, size_t total) {
    // dst[phase][q][n][j], k = q*32 + j = tap*cin_pad + c
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = idx & 31;
        size_t r = idx >> 5;
        const int n = r % p.cout_pad;
        r /= p.cout_pad;
        const int q = r % nq;
        const int phase = r / nq;
        const int k = q * 32 + j;
        int tap = k / p.cin_pad, c = k % p.cin_pad;
        if (p.k_order == 1) {      // chunk-major: q = cc * ntap + tap
            const int ntap = p.KH * p.KW;
            tap = q % ntap;
            c = (q / ntap) * 32 + j;
        }
        float v = 0.f;
        if (n < p.Cout && c < p.Cin && tap < p.KH * p.KW) {
            const int kh = tap / p.KW, kw = tap % p.KW;
            if (p.transposed) {
                // ConvTranspose2d(4,2,1) weight [Cin][Cout][4][4]; output phase (a,b) uses ky = a ? 2-2kh : 3-2kh
                const int a = phase >> 1, bph = phase & 1;
                const int ky = a ? 2 - 2 * kh : 3 - 2 * kh, kx = bph ? 2 - 2 * kw : 3 - 2 * kw;
                v = p.w[(((size_t)c * p.Cout + n) * 4 + ky) * 4 + kx];
            } else {
                v = p.w[(((size_t)n * p.Cin + c) * p.KH + kh) * p.KW + kw];
            }
            if (p.gamma) v *= p.gamma[c];
        }
        reinterpret_cast<T*>(p.dst)[idx] = from_f32<T>(v);
    }
}

__global__ void fold_tables_kernel(const float* w, const float* bias, const float* gamma, const float* beta, int Cout,
                                   int Cin, int KH, int KW, float* t1, float* t2) {
    // one wave per (cls, o)
    const int ncls = (KH == 3 && KW == 3) ? 9 : 1;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= ncls * Cout) return;
    const int cls = wid / Cout, o = wid % Cout;
    const int ch = cls / 3, cw = cls % 3;
    float a1 = 0.f, a2 = 0.f;
    for (int idx = lane; idx < Cin * KH * KW; idx += 64) {
        const int c = idx / (KH * KW), t = idx % (KH * KW), kh = t / KW, kw = t % KW;
        bool ok = true;
        if (ncls == 9) ok = !(ch == 0 && kh == 0) && !(ch == 2 && kh == 2) && !(cw == 0 && kw == 0) && !(cw == 2 && kw == 2);
        if (ok) {
            const float wv = w[((size_t)o * Cin + c) * KH * KW + t];
            a1 += wv * beta[c];
            a2 += wv * gamma[c];
        }
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
        t1[cls * Cout + o] = a1 + (bias ? bias[o] : 0.f);
        t2[cls * Cout + o] = a2;
    }
}

}  // namespace

extern "C" size_t ds_pack_conv_elems(int cin_pad, int KH, int KW, int cout_pad, int transposed) {
    const size_t nq = ((size_t)KH * KW * cin_pad + 31) / 32;
    return (transposed ? 4 : 1) * nq * cout_pad * 32;
}

extern "C" int ds_pack_conv_weight(const ds_pack_conv_params* p, void* stream) {
    DS_REQUIRE(p && p->w && p->dst, "pack_conv: null pointer");
    DS_REQUIRE(p->cin_pad >= p->Cin && p->cout_pad >= p->Cout, "pack_conv: pads too small");
    DS_REQUIRE(!p->transposed || (p->KH == 2 && p->KW == 2), "pack_conv: transposed packs 2x2 sub-kernels");
    DS_REQUIRE(p->k_order == 0 || (p->k_order == 1 && p->cin_pad % 32 == 0 && !p->transposed), "pack_conv: k_order 1 needs cin_pad %% 32 == 0");
    const int nq = (p->KH * p->KW * p->cin_pad + 31) / 32;
    const size_t total = ds_pack_conv_elems(p->cin_pad, p->KH, p->KW, p->cout_pad, p->transposed);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == DS_BF16) hipLaunchKernelGGL(pack_conv_kernel<bf16>, dim3(blocks), dim3(256), 0, st, *p, nq, total);
    else hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(blocks), dim3(256), 0, st, *p, nq, total);
    DS_CHECK_LAUNCH("pack_conv");
    return DS_OK;
}

extern "C" int ds_conv_fold_tables(const float* w, const float* bias, const float* gamma, const float* beta, int Cout,
                                   int Cin, int KH, int KW, float* t1, float* t2, void* stream) {
    DS_REQUIRE(w && gamma && beta && t1 && t2, "fold_tables: null pointer");
    DS_REQUIRE((KH == 3 && KW == 3) || (KH == 1 && KW == 1), "fold_tables: 3x3 or 1x1 only");
    const int ncls = KH == 3 ? 9 : 1;
    const int waves = ncls * Cout;
    hipLaunchKernelGGL(fold_tables_kernel, dim3((waves + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, bias,
                       gamma, beta, Cout, Cin, KH, KW, t1, t2);
    DS_CHECK_LAUNCH("fold_tables");
    return DS_OK;
}
