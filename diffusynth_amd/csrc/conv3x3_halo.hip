// 3x3 stride-1 pad-1 convolution, bf16 MFMA, direct from an LDS halo tile (gfx950).
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers the im2col A tile for each of the 9
// taps: 9x the global->LDS traffic and 9x the address arithmetic for the same input pixels.  Here a
// block owns a TH x TW patch of output pixels (TH*TW = 256) of one sample; per 32-channel chunk it
// stages the (TH+2) x (TW+2) input halo ONCE in LDS (rows of 64 B padded to 80 B: conflict-free
// ds_read_b128 with immediate tap offsets) and the 9 taps read their fragments straight from it at
// shifted pixel offsets — no im2col tile is ever materialised.  Only the BN x 32 weight tile is
// streamed per (chunk, tap): 3-slot register ring -> 3-buffer LDS ring.
//   K order: channel-chunk major, tap minor; weights stay in the generic packed layout
//            [tap*NCC + cc][cout_pad][32], so one packing serves both kernels.
//   MFMA 32x32x16 bf16 issued as mfma(W, X): a lane's accumulators are ONE pixel x 16 channels, which
//   makes the epilogue register-only (conv_epilogue.hpp: conv_epilogue_t).
//   Default variant <256, 96, 4 x 1 waves, TW <= 32>: wave tile 64 px x 96 ch, 77 KB LDS => two independent
//   blocks per CU; the 8-wave 256 x 192 / 256 x 96 and the 128-pixel variants remain for A/B runs.
//   global->LDS bytes per MFMA flop: ~2.6x lower than the im2col kernel.
#include <type_traits>

#include "common.hpp"
#ifndef DS_ABLATE
#define DS_ABLATE 0   // diagnostic builds only (tools/ablate.sh): bit0 no weight stream, bit1 no halo refill, bit2 no barrier, bit3 no fragment reads, bit4 no epilogue, bit5 no K loop
#endif
#include "conv_epilogue.hpp"
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif
#ifndef DS_STAMP
#define DS_STAMP 0   // diagnostic build: per-wave s_memtime stamps around the K loop's barriers -> p.slab (8 longs per wave)
#endif

namespace {

// halo bytes for a BM-pixel patch: max over TW in {64,32,16,8} of (BM/TW + 2) * (TW + 2) * 64, rounded up
// Pixel / weight rows are padded from 64 to 80 bytes in LDS: 5 x 16 B is coprime with the 16 slots of a 256-B
// bank row, so any 16 lanes reading distinct consecutive rows are conflict-free WITHOUT an XOR swizzle, and
// every tap / k-substep offset becomes an immediate of the ds_read (no address arithmetic in the K loop).
constexpr int PSTR = 80;
// TWMAX = widest patch the variant uses: 64 -> 6 x 66 pixels (256-pixel patch), 32 -> 10 x 34
constexpr int halo_bytes(int BM, int TWL_MAX = 6) { return BM == 256 ? (TWL_MAX == 6 ? 6 * 66 : 10 * 34) * PSTR : 4 * 66 * PSTR; }

__device__ __forceinline__ int swz64(int row, int chunk) { return row * PSTR + (chunk << 4); }

// Which pixel of a patch row an MFMA lane owns is free (the epilogue uses the same map).  With 16- and 8-pixel rows a
// 32-pixel fragment spans 2 / 4 patch rows whose LDS offsets (row pitch TW + 2 pixels of 80 B) collide inside the 16-lane
// groups of ds_read_b128; rotating the columns of row r by this amount makes the 16-wide case conflict-free and cuts the
// 8-wide case from 12 to 4 colliding lanes per 32 (offline search over all rotations).
__device__ __forceinline__ int lane_rot(int row, int twl) { return twl == 4 ? (row & 1) * 14 : (twl == 3 ? (row & 3) * 2 : 0); }

template <int BM, int BN, int WM, int WN, int OCC, int TWL_MAX = 6>
__global__ __launch_bounds__(WM* WN * 64, OCC) void conv3x3_halo_kernel(const ds_conv_params p, int twl) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int HALO_BYTES = halo_bytes(BM, TWL_MAX);
    constexpr int B_BYTES = BN * PSTR;
    constexpr int B_IT = (BN * 4 + NT - 1) / NT;
    constexpr int H_IT = (HALO_BYTES / PSTR * 4 + NT - 1) / NT;
    static_assert((BM == 256 || BM == 128) && TM % 32 == 0 && TN % 32 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsH = smem;                   // [2][HALO_BYTES]
    char* const ldsB = smem + 2 * HALO_BYTES;  // [3][BN][32] bf16 (ring: read s, prefetch-read s+1, write s+2)
    float* const red = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long st_k0 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz wall clock at kernel entry
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fh = lane >> 5;
    const int TW = 1 << twl, TH = BM >> twl, HC = TW + 2, npx = (TH + 2) * HC;
    const int tiles_w = (p.W + TW - 1) >> twl;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int b = blockIdx.z / ksplit, kz = blockIdx.z - b * ksplit, n0 = blockIdx.y * BN;
    const int Cin = p.C0, NCC_all = Cin >> 5;
    const int NCC = NCC_all / ksplit, cc_lo = kz * NCC;   // this block's K slice: channel chunks [cc_lo, cc_lo + NCC)

    float gn_a = 1.f, gn_am = 0.f;

    const bf16* src = reinterpret_cast<const bf16*>(p.src0) + (size_t)b * p.H * p.W * Cin;
    // (a thread whose first slot lies past a narrow BN tile fetches slot 0: its loads are unconditional but never stored,
    // and must not run past the end of the packed weights on the last K chunk of the last channel tile)
    const bf16* wbase = reinterpret_cast<const bf16*>(p.wpk) + (size_t)n0 * 32 + (tid * 8 < BN * 32 ? tid * 8 : 0);
    const size_t wstride = (size_t)p.cout_pad * 32;  // elements per packed K chunk

    // ---- halo loader: slot = tid + it*512 -> (pixel, 16-B chunk); fixed per thread for the whole kernel
    int hoff[H_IT];   // global element offset of the slot's 16 bytes, -1 = zero fill (outside the image), -2 = no slot
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
        const int slot = tid + it * NT, px = slot >> 2, ch = slot & 3;
        hoff[it] = -2;
        if (px < npx) {
            const int hr = px / HC, hc = px - hr * HC;
            const int hi = h0 + hr - 1, wi = w0 + hc - 1;
            hoff[it] = ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? (hi * p.W + wi) * Cin + ch * 8 : -1;
        }
    }
    // B tiles are prefetched PF steps ahead through a register ring: the weights of the big layers (10 MB)
    // come from Infinity Cache / HBM at ~1 us under load, far longer than one step of MFMA work.
    constexpr int PF = 3;
    u32x4 rh[(H_IT + 1) / 2], rb[PF][B_IT];
    // the halo of the next chunk is fetched in two halves (slots [0,HH) and [HH,H_IT)) to keep few registers live
    constexpr int HH = (H_IT + 1) / 2;
    // NOTE: every global load in the K loop is UNCONDITIONAL (clamped address + select).  A load under a branch
    // makes hipcc fall back to s_waitcnt vmcnt(0) at the merge point, which drains the whole prefetch ring each step.
    auto load_halo = [&](int cc, int lo, int hi_) {
#pragma unroll
        for (int it = 0; it < H_IT; ++it)
            if (it >= lo && it < hi_) {
                const u32x4 v = DS_LD(u32x4, src + (hoff[it] >= 0 ? hoff[it] : 0) + (cc_lo + cc) * 32, DS_BX_SRC0);
                rh[it - lo] = hoff[it] >= 0 ? v : u32x4{0u, 0u, 0u, 0u};
            }
    };
    auto store_halo = [&](int buf, int lo, int hi_) {
        char* h = ldsH + buf * HALO_BYTES;
#pragma unroll
        for (int it = 0; it < H_IT; ++it)
            if (it >= lo && it < hi_ && hoff[it] != -2) {
                const int slot = tid + it * NT;
                *reinterpret_cast<u32x4*>(h + swz64(slot >> 2, slot & 3)) = rh[it - lo];
            }
    };
    const int nsteps = NCC * 9;
    int boff[B_IT];   // element offset of this thread's slot inside a weight tile (0 for slots past the tile: dummy load)
#pragma unroll
    for (int it = 0; it < B_IT; ++it) boff[it] = ((it + 1) * NT <= BN * 4 || tid + it * NT < BN * 4) ? it * NT * 8 : 0;
    auto load_b = [&](auto slotc, int s) {   // s = global step index = cc*9 + tap (clamped: tail loads are dummies)
        constexpr int slot_ = decltype(slotc)::value;
        s = s < nsteps ? s : nsteps - 1;
        const int cc = s / 9, tap = s - cc * 9;
        const bf16* w = wbase + (size_t)(tap * NCC_all + cc_lo + cc) * wstride;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[slot_][it] = DS_LD(u32x4, w + boff[it], DS_BX_W);
    };
    auto store_b = [&](auto slotc, int buf) {
        constexpr int slot_ = decltype(slotc)::value;
        char* bb = ldsB + buf * B_BYTES;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int slot = tid + it * NT;
            if ((it + 1) * NT <= BN * 4 || slot < BN * 4) *reinterpret_cast<u32x4*>(bb + swz64(slot >> 2, slot & 3)) = rb[slot_][it];
        }
    };

    // ---- per-lane fragment bases
    int p0[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * TM + i * 32 + frow;
        p0[i] = ((ml >> twl) * HC + ((ml + lane_rot(ml >> twl, twl)) & (TW - 1))) * PSTR + fh * 16;   // this lane's pixel (+ k half)
    }
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment double buffer: set 0 = k-substep 0, set 1 = k-substep 1 of a step.  While the six MFMAs of one
    // substep execute, the ds_reads of the next substep (possibly of the next step) are already in flight, so
    // the LDS-read phase and the MFMA phase of the barrier-synchronised waves overlap instead of alternating.
    bf16x8 fa[2][FM], fb[2][FN];
    int bofs[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) bofs[j] = (wn * TN + j * 32 + frow) * PSTR + fh * 16;
    auto read_frags = [&](auto setc, int hbuf, int bbuf, int shift, int sub) {
        constexpr int set = decltype(setc)::value;
        const char* h = ldsH + hbuf * HALO_BYTES + shift * PSTR + sub * 32;
        const char* bb = ldsB + bbuf * B_BYTES + sub * 32;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(h + p0[i]);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(bb + bofs[j]);
    };
    auto mma = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
#if (DS_ABLATE & 128)
        // timing experiment only (wrong results): same FLOPs issued as 16x16x32 MFMAs on the same registers
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                f32x4 c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i], fb[set][j], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i], fb[set][j], c1, 0, 0, 0);
                acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
                acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
            }
#else
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);   // D^T = W . X^T
#endif
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // ---- main loop: channel chunks x 9 taps, one barrier per step.
    // LDS: weights in a 3-buffer ring (step s computes from buffer s%3, prefetch-reads (s+1)%3, writes (s+2)%3),
    // halo double-buffered per chunk.  Global weight tiles travel through a 3-slot register ring, 3 steps ahead.
    // 9 % 3 == 0, so every ring index is a compile-time constant inside the unrolled tap loop.
    static_assert(PF == 3, "ring arithmetic below assumes 3");
    load_halo(0, 0, HH);
    store_halo(0, 0, HH);
    load_halo(0, HH, H_IT);
    store_halo(0, HH, H_IT);
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    load_b(R0{}, 0);
    load_b(R1{}, 1);
    store_b(R0{}, 0);
    store_b(R1{}, 1);
    load_b(R2{}, 2);
    load_b(R0{}, 3);
    load_b(R1{}, 4);
    // GroupNorm statistics of the input from the producer's partials, overlapped with the prologue loads
    if (p.gn_part && ksplit == 1) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, gn_a, gn_am);
    __syncthreads();
    read_frags(S0{}, 0, 0, 0, 0);
    long st_lgkm = 0, st_bar = 0;
    const long st_t0 = DS_STAMP ? __builtin_amdgcn_s_memtime() : 0;
    const long st_r0 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int cc = 0; cc < ((DS_ABLATE & 32) ? 0 : NCC); ++cc) {
        const int ccn = cc + 1 < NCC ? cc + 1 : cc;
        auto step = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int rs = (tap + 2) % 3;
            const int s = cc * 9 + tap;
            if constexpr (!(DS_ABLATE & 8)) read_frags(S1{}, cc & 1, tap % 3, (tap / 3) * HC + (tap % 3), 1);
            // branch-free body: tail iterations load clamped dummies and store into buffers nobody reads
            if constexpr (!(DS_ABLATE & 1)) {
                store_b(std::integral_constant<int, rs>{}, rs);
                load_b(std::integral_constant<int, rs>{}, s + 5);
            }
            if constexpr (!(DS_ABLATE & 2)) {
                if constexpr (tap == 1) load_halo(ccn, 0, HH);
                if constexpr (tap == 4) load_halo(ccn, HH, H_IT);
            }
            mma(S0{});
            if constexpr (!(DS_ABLATE & 8)) {
                if constexpr (tap < 8) read_frags(S0{}, cc & 1, (tap + 1) % 3, ((tap + 1) / 3) * HC + ((tap + 1) % 3), 0);
                else read_frags(S0{}, (cc + 1) & 1, 0, 0, 0);
            }
            mma(S1{});
            if constexpr (!(DS_ABLATE & 2)) {
                if constexpr (tap == 3) store_halo((cc + 1) & 1, 0, HH);
                if constexpr (tap == 7) store_halo((cc + 1) & 1, HH, H_IT);
            }
            if constexpr (DS_STAMP) {
                const long ta = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const long tb = __builtin_amdgcn_s_memtime();
                __syncthreads();
                const long tc = __builtin_amdgcn_s_memtime();
                st_lgkm += tb - ta;
                st_bar += tc - tb;
            } else if constexpr (!(DS_ABLATE & 4)) __syncthreads();
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
    }

    if constexpr (DS_STAMP) {
        const long st_t1 = __builtin_amdgcn_s_memtime();
        if (p.slab && ksplit == 1 && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
            d[0] = st_t1 - st_t0; d[1] = st_lgkm; d[2] = st_bar; d[3] = nsteps;
            d[4] = st_r0 - st_k0; d[5] = __builtin_amdgcn_s_memrealtime() - st_r0; d[6] = st_k0;
        }
    }
    // ---- epilogue (conv_epilogue.hpp)
    auto coord = [&](int ml) {
        ConvCoord c;
        c.ho = h0 + (ml >> twl);
        c.wo = w0 + ((ml + lane_rot(ml >> twl, twl)) & (TW - 1));
        c.ok = c.ho < p.H && c.wo < p.W;
        c.pix = c.ho * p.W + c.wo;
        return c;
    };
    float s1 = 0.f, s2 = 0.f;
    float* const shl = reinterpret_cast<float*>(smem);   // [ncls][BN] shift table (the K loop ended on a barrier: LDS is free)
    if constexpr ((DS_ABLATE & 16) != 0) {   // timing experiment: no epilogue (the accumulators stay live through a never-true store)
        float t = 0.f;
        for (int i = 0; i < FM; ++i)
            for (int j = 0; j < FN; ++j)
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) shl[0] = t;
        return;
    }
    if (ksplit > 1) {
        // raw fp32 partial sums of this K slice -> slab[kz][b]; bias / fold / activation / residual / statistics
        // happen in ds_conv_splitk_reduce
        ds_conv_params q = p;
        q.out = p.slab;
        q.out_C = (p.Cout + 7) / 8 * 8;
        q.out_c0 = 0;
        conv_epilogue_t_body<float, FM, FN, BN, DS_ACT_NONE, false, true>(q, acc, kz * p.B + b, n0, wn * TN, wm * TM, p.H * p.W, shl, coord, s1, s2, 1.f);
        return;
    }
    if (!p.gn_part && p.gn_ab) {
        gn_a = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
        gn_am = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
    }
    conv_shift_table<BN>(p, n0, gn_am, shl);
    __syncthreads();
    const long st_e1 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    conv_epilogue_t<bf16, FM, FN, BN>(p, acc, b, n0, wn * TN, wm * TM, p.H * p.W, shl, coord, s1, s2, gn_a);
    const long st_e2 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    __syncthreads();
    if (p.stats_part) {
        const int parts = gridDim.x * gridDim.y;
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2);
    }
    if constexpr (DS_STAMP) {
        if (p.slab && ksplit == 1 && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
            d[7] = __builtin_amdgcn_s_memrealtime() - st_k0;
            d[1] = st_e1 - st_k0;      // (overwrites the lgkm counter) end of shift-table phase
            d[2] = st_e2 - st_k0;      // (overwrites the barrier counter) end of the epilogue body
        }
    }
}

int halo_twl(int W, int twl_max = 6) {
    // patch width: smallest power of two >= min(W, 2^twl_max), at least 8
    int twl = 3;
    while ((1 << twl) < W && twl < twl_max) ++twl;
    return twl;
}

template <int BM, int BN, int WM, int WN, int OCC = 2, int TWL_MAX = 6>
int launch_halo(const ds_conv_params& p, hipStream_t st) {
    constexpr int NW = WM * WN;
    constexpr size_t lds_main = 2 * (size_t)halo_bytes(BM, TWL_MAX) + 3 * (size_t)BN * PSTR;
    constexpr size_t lds_epi = NW * 32 * (size_t)(BN / WN + 4) * sizeof(float);
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    auto kern = conv3x3_halo_kernel<BM, BN, WM, WN, OCC, TWL_MAX>;
    DS_SET_MAX_LDS(kern, lds, "conv3x3_halo");
    const int twl = halo_twl(p.W, TWL_MAX), TW = 1 << twl, TH = BM >> twl;
    dim3 grid(((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW), p.cout_pad / BN, p.B * (p.ksplit > 1 ? p.ksplit : 1));
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_HALO);
        ds_conv_bounds_table(p, DS_K_CONV_HALO, grid.x * grid.y, &h.t);
        h.publish(st);
    }
#endif
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, p, twl);
    DS_CHECK_LAUNCH("conv3x3_halo");
    return DS_OK;
}

static void halo_dims(int tile, int* bm, int* bn) {
    *bm = (tile == DS_CONV_TILE_HALO_128x192 || tile == DS_CONV_TILE_HALO_128x96) ? 128 : 256;
    *bn = (tile == DS_CONV_TILE_HALO_256x192 || tile == DS_CONV_TILE_HALO_128x192 || tile == DS_CONV_TILE_HALO_256x192_W4) ? 192 : 96;
}

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
    if (i < (long)HW * cv8) {
        const int pix = i / cv8, n = (i - (long)pix * cv8) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // all KS slab reads in flight at once: with a rolled loop each pair of loads waited for the previous one
        f32x4 sa[KS], sc[KS];
#pragma unroll
        for (int z = 0; z < KS; ++z) {
            const float* sp = p.slab + (((size_t)z * p.B + b) * HW + pix) * cs + n;
            sa[z] = DS_LD(f32x4, sp, DS_BX_AUX0);
            sc[z] = DS_LD(f32x4, sp + 4, DS_BX_AUX0);
        }
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
        const size_t o = ((size_t)b * HW + pix) * p.out_C + p.out_c0 + n;
        if (p.res) {
            float rv[8];
            vec16_load<bf16>(reinterpret_cast<const bf16*>(p.res) + o, rv, DS_BX_RES);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += rv[q];
        }
        vec16_store<bf16>(reinterpret_cast<bf16*>(p.out) + o, v, DS_BX_OUT);
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
    DS_REQUIRE(p && (p->ksplit == 2 || p->ksplit == 4 || p->ksplit == 8) && p->slab && p->out, "splitk_reduce: needs ksplit in {2, 4, 8}, slab and out");
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
        h.publish(st);
    }
#endif
    if (p->ksplit == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, grid, dim3(RED_BLOCK), 0, st, *p);
    else if (p->ksplit == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, grid, dim3(RED_BLOCK), 0, st, *p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<8>, grid, dim3(RED_BLOCK), 0, st, *p);
    DS_CHECK_LAUNCH("splitk_reduce");
    return DS_OK;
}

namespace {
void halo_dims2(int tile, int* bm, int* bn) { halo_dims(tile, bm, bn); }
}  // namespace

// called from ds_conv_igemm for tile ids DS_CONV_TILE_HALO_*
int ds_conv3x3_halo_parts(const ds_conv_params* p) {
    if (p->ksplit > 1) return (int)(((long)p->Ho * p->Wo * (p->transposed ? 4 : 1) * ((p->Cout + 7) / 8) + RED_BLOCK - 1) / RED_BLOCK);
    int bm, bn;
    halo_dims(p->tile, &bm, &bn);
    const int twl = halo_twl(p->W, (p->tile == DS_CONV_TILE_HALO_256x96_W4 || p->tile == DS_CONV_TILE_HALO2_256x96 || p->tile == DS_CONV_TILE_HALO3_256x96) ? 5 : 6), TW = 1 << twl, TH = bm >> twl;
    return ((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW) * (p->cout_pad / bn);
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_halo(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

int ds_conv3x3_halo_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv3x3_halo: bf16 only");
    DS_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad_h == 1 && p->pad_w == 1 && !p->transposed,
               "conv3x3_halo: 3x3 stride 1 pad 1 only");
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0, "conv3x3_halo: single source, Cin multiple of 32 (got %d+%d)", p->C0, p->C1);
    DS_REQUIRE(p->Ho == p->H && p->Wo == p->W && !p->out_nchw_f32, "conv3x3_halo: same-size NHWC output only");
    DS_REQUIRE(p->ksplit <= 1 || (p->slab && (p->C0 / 32) % p->ksplit == 0),
               "conv3x3_halo: ksplit=%d needs a slab and must divide the %d channel chunks", p->ksplit, p->C0 / 32);
    switch (p->tile) {
        case DS_CONV_TILE_HALO_256x192: return launch_halo<256, 192, 4, 2>(*p, st);
        case DS_CONV_TILE_HALO_256x96: return launch_halo<256, 96, 8, 1>(*p, st);
        case DS_CONV_TILE_HALO_256x192_W4: return launch_halo<256, 192, 2, 2, 1>(*p, st);
        case DS_CONV_TILE_HALO_256x96_W4: return launch_halo<256, 96, 4, 1, 2, 5>(*p, st);
        case DS_CONV_TILE_HALO_128x192: return launch_halo<128, 192, 2, 2>(*p, st);
        default: return launch_halo<128, 96, 4, 1>(*p, st);
    }
}
