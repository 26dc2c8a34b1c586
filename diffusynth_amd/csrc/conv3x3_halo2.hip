// 3x3 stride-1 pad-1 convolution, bf16 MFMA from an LDS input halo — hand-scheduled K loop (gfx950).
//
// Same tile and data flow as conv3x3_halo.hip's <256 px x 96 ch, 4 waves> variant (two blocks per CU): per
// 32-channel chunk the (TH+2) x (TW+2) input halo is staged once in LDS, the nine taps read their pixel fragments
// from it at shifted offsets, only the 96 x 32 weight tile streams per (chunk, tap) through a 3-slot register
// ring into a 3-buffer LDS ring; mfma(W, X) keeps one pixel per lane for the register-only epilogue.
//
// What changed, and why (measured on MI355X with in-kernel stamps): ONE wave alone on its SIMD needed 838 cycles
// per tap step for 384 cycles of MFMA issue — the in-order wave spent the rest issuing ~100 scalar / vector /
// LDS / VMEM instructions that hipcc had gathered into clumps between groups of MFMAs (64-bit address
// arithmetic with scalar multiplies per weight tile, selects for the clamped loads, exec-masked branches around
// partial stores), and in short waits.  Two co-resident waves hid part of it (1100 cycles per step pair for 768 of
// MFMA = 70 %).  Here the step is built so that almost nothing but MFMAs, LDS reads and a handful of loads remain,
// and they are placed one or two per MFMA gap (an MFMA holds the SIMD's issue for 8 of its 32 cycles):
//   * weights are packed CHUNK-MAJOR ([cc*9 + tap][cout_pad][32], ds_pack_conv_params.k_order = 1): the tile of
//     step s+1 is the tile of step s plus one constant stride;
//   * every global load is a buffer load (resource descriptor + per-thread 32-bit offset fixed for the whole kernel
//     + ONE scalar offset that advances per step / per chunk): no vector address arithmetic in the loop, and the
//     range check returns zeros for out-of-image halo pixels and for idle slots (no select, no branch);
//   * every LDS address is a per-thread base plus an immediate (the chunk loop is unrolled by two for the halo
//     double buffer); idle lanes of the partial last store iteration write into a pad column instead of branching;
//   * the order inside a step is pinned with sched_group_barrier (one LDS / VMEM instruction per MFMA gap), the
//     barrier is a raw s_barrier behind a COUNTED lgkmcnt that only waits for this step's LDS writes — the fragment
//     reads of the next step stay in flight across it;
//   * the border-class shift table is built in the prologue (its two dependent global reads used to sit between the
//     K loop and the epilogue, 1.8 us per block), the prologue issues all its loads in one round trip, and the
//     epilogue fetches all residual vectors before it touches the accumulators.
#include <type_traits>

#include "common.hpp"
#include "conv_epilogue.hpp"
#ifndef DS_STAMP
#define DS_STAMP 0   // diagnostic build: per-wave s_memtime stamps around the K loop's waits -> p.slab (8 longs per wave), tools/conv_microbench.py --stamp 1
#endif

int ds_conv3x3_halo_parts(const ds_conv_params* p);   // conv3x3_halo.hip (same grid / partial layout)
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif

namespace {

constexpr int PSTR = 80;                       // LDS row pitch: 64 B of data + 16 B pad (conflict-free ds_read_b128)
constexpr int BM = 256, BN = 96, NW = 4, NT = 256;
constexpr int FM = 2, FN = 3;                  // wave tile 64 px x 96 ch = 2 x 3 accumulators of 32 x 32
constexpr int HALO_PX = 340;                   // max over TW in {32, 16, 8} of (256/TW + 2) * (TW + 2)
constexpr int HALO_BYTES = HALO_PX * PSTR;     // 27200
constexpr int B_BYTES = BN * PSTR;             // 7680
constexpr int SHL_BYTES = 9 * BN * 4;          // 3456: shift table [9 border classes][BN]
constexpr int OFF_B = 0, OFF_SHL = 3 * B_BYTES, OFF_H = OFF_SHL + SHL_BYTES;
constexpr int LDS_BYTES = OFF_H + 2 * HALO_BYTES;   // 80896 <= 81920: two blocks per CU
constexpr int H_IT = (HALO_PX * 4 + NT - 1) / NT;   // 6 (the last iteration is partial)
constexpr int HH = H_IT / 2;                        // halo refill in two halves of 3 loads
constexpr unsigned VOFF_NONE = 0x80000000u;         // beyond any num_records: the buffer range check returns zeros

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// 16-byte buffer load: address = base + soff + voff; voff >= num_records (VOFF_NONE) reads zeros.
__device__ __forceinline__ u32x4 buf_ld16(rsrc_t rs, const char* base, unsigned voff, unsigned soff, int bounds_buf) {
#if DS_BOUNDS
    if (voff < VOFF_NONE && !ds_bx_ok(base + soff + voff, bounds_buf, 16)) return u32x4{0u, 0u, 0u, 0u};
#endif
    (void)base; (void)bounds_buf;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
}

__device__ __forceinline__ int lane_rot(int row, int twl) { return twl == 4 ? (row & 1) * 14 : (twl == 3 ? (row & 3) * 2 : 0); }

// LLVM SchedGroupMask bits
constexpr int SG_MFMA = 0x8, SG_VMEM = 0x10, SG_DSR = 0x100, SG_DSW = 0x200;

__global__ __launch_bounds__(NT, 2) void conv3x3_halo2_kernel(const ds_conv_params p, int twl, int hc_magic) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const shl = reinterpret_cast<float*>(smem + OFF_SHL);
    float* const red = reinterpret_cast<float*>(smem + OFF_B);      // reused after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long st_k0 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    const int frow = lane & 31, fh = lane >> 5;
    const int TW = 1 << twl, TH = BM >> twl, HC = TW + 2, npx = (TH + 2) * HC;
    const int tiles_w = (p.W + TW - 1) >> twl;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int b = blockIdx.z / ksplit, kz = blockIdx.z - b * ksplit, n0 = blockIdx.y * BN;
    const int Cin = p.C0, NCC_all = Cin >> 5;
    const int NCC = NCC_all / ksplit, cc_lo = kz * NCC;
    const int nsteps = NCC * 9;

    // ---- resource descriptors (wave-uniform) and per-thread offsets, all fixed for the whole kernel
    // fused res_conv (K steps appended to the nine-tap chunks; whole-K launches only)
    const int NR = ksplit == 1 ? p.res_steps : 0, R0 = p.res_C0 >> 5;
    const char* const wbase = reinterpret_cast<const char*>(p.wpk);
    const unsigned wbytes = (unsigned)(NCC_all * 9 + p.res_steps) * p.cout_pad * 64;
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), (short)0, (int)wbytes, 0x00020000);

    // Linear chunk order of a block: first the NR one-step chunks of the fused res_conv (32 channels of res_src0, then of
    // res_src1 placed at its pad offset: pad_and_concat, components:210-249), then the NCC nine-tap chunks of the 3x3 input.
    // (res first: its extra state — six staging registers, source descriptors — is dead before the hot loop starts, so the
    // 3x3 loop keeps its register budget; a res phase AFTER the loop spilled 147 registers into it.)
    // hvo = byte offset of this thread's 16 B of chunk 0 inside the current source's sample (VOFF_NONE: outside the image or
    // not needed — the range check returns zeros without touching memory); rebuilt where the source changes (uniform branch).
    unsigned hvo[H_IT];
    rsrc_t rs_h;
    const char* hbase;
    int h_buf = DS_BX_SRC0;
    auto use_source = [&](const void* ptr, int sH, int sW, int sC, int offh, int offw, bool centre_only, int bounds_buf) {
        hbase = reinterpret_cast<const char*>(ptr) + (size_t)b * sH * sW * sC * 2;
        rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(hbase), (short)0, (int)((unsigned)sH * sW * sC * 2), 0x00020000);
        h_buf = bounds_buf;
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const int slot = tid + it * NT, px = slot >> 2, ch = slot & 3;
            hvo[it] = VOFF_NONE;
            if (px < npx) {
                const int hr = (px * hc_magic) >> 16, hc = px - hr * HC;      // px / HC without the 40-instruction division (exact for px < 1500)
                const int hi = h0 + hr - 1 - offh, wi = w0 + hc - 1 - offw;
                const bool need = !centre_only || (hr >= 1 && hr <= TH && hc >= 1 && hc <= TW);   // a 1x1 never reads the halo ring
                if (need && (unsigned)hi < (unsigned)sH && (unsigned)wi < (unsigned)sW) hvo[it] = (unsigned)((hi * sW + wi) * sC + ch * 8) * 2u;
            }
        }
    };
    unsigned h_so = 0;           // scalar byte offset of the chunk inside a pixel's channels
    // select res chunk r (0 <= r < NR) or, for r == NR, the first 3x3 chunk; called with consecutive r only
    auto set_res_src = [&](int r) {
        if (r == R0 && r < NR) use_source(p.res_src1, p.res_H1, p.res_W1, p.res_C1, p.res_off_h1, p.res_off_w1, true, DS_BX_AUX1);
        if (r == NR) use_source(p.src0, p.H, p.W, Cin, 0, 0, false, DS_BX_SRC0);
        h_so = r == NR ? (unsigned)cc_lo * 64u : (unsigned)(r < R0 ? r : r - R0) * 64u;
    };
    if (NR > 0) use_source(p.res_src0, p.H, p.W, p.res_C0, 0, 0, true, DS_BX_AUX0);
    else use_source(p.src0, p.H, p.W, Cin, 0, 0, false, DS_BX_SRC0);
    // LDS store offsets: slot -> row (slot >> 2) * 80 + chunk (slot & 3) * 16 is linear in `it` (+ 64 rows * 80 B);
    // lanes without a slot in the partial last iteration write the 16-byte pad column of row 0 instead
    const int lds0 = (tid >> 2) * PSTR + (tid & 3) * 16;
    const int hst_last = (tid + (H_IT - 1) * NT) < npx * 4 ? lds0 + (H_IT - 1) * 64 * PSTR : 64;
    const int bst1 = tid < BN * 4 - NT ? lds0 + 64 * PSTR : 64;
    const unsigned wvo0 = (unsigned)tid * 16u, wvo1 = tid < BN * 4 - NT ? (unsigned)(tid + NT) * 16u : VOFF_NONE;
    const unsigned wstep = (unsigned)p.cout_pad * 64u;                                  // bytes per K step
    // wpk = [res tiles (p.res_steps)][3x3 tiles]: the stream of a fused launch starts at the res tiles, any other launch skips them
    const unsigned w_first = ((unsigned)(p.res_steps - NR + cc_lo * 9) * p.cout_pad + n0) * 64u, w_last = w_first + (unsigned)(nsteps + NR - 1) * wstep;
    unsigned w_pf = w_first;     // scalar offset of the next weight tile to fetch (clamped at the last real step: tail loads are dummies)

    u32x4 rb[3][2], rh[H_IT];     // rh: the 3x3 chunks refill the halo in two halves through rh[0 .. HH); the 1-step res chunks use all six
    auto load_b = [&](auto slotc) {
        constexpr int sl = decltype(slotc)::value;
        rb[sl][0] = buf_ld16(rs_w, wbase, wvo0, w_pf, DS_BX_W);
        rb[sl][1] = buf_ld16(rs_w, wbase, wvo1, w_pf, DS_BX_W);
        const unsigned nx = w_pf + wstep;
        w_pf = nx < w_last ? nx : w_last;
    };
    auto store_b = [&](auto slotc, auto bufc) {
        constexpr int sl = decltype(slotc)::value, buf = decltype(bufc)::value;
        *reinterpret_cast<u32x4*>(smem + OFF_B + buf * B_BYTES + lds0) = rb[sl][0];
        *reinterpret_cast<u32x4*>(smem + OFF_B + buf * B_BYTES + bst1) = rb[sl][1];
    };
    // (the source / chunk of a halo load is whatever was selected last: rs_h / hvo / h_so)
    auto load_halo_to = [&](u32x4* dst, auto halfc) {
        constexpr int half = decltype(halfc)::value;
#pragma unroll
        for (int k = 0; k < HH; ++k) dst[k] = buf_ld16(rs_h, hbase, hvo[half * HH + k], h_so, h_buf);
    };
    auto load_halo = [&](auto halfc) { load_halo_to(rh, halfc); };
    auto store_halo = [&](auto bufc, auto halfc) {
        constexpr int buf = decltype(bufc)::value, half = decltype(halfc)::value;
        char* const h = smem + OFF_H + buf * HALO_BYTES;
#pragma unroll
        for (int k = 0; k < HH; ++k) {
            const int it = half * HH + k;
            if (it == H_IT - 1) *reinterpret_cast<u32x4*>(h + hst_last) = rh[k];
            else *reinterpret_cast<u32x4*>(h + lds0 + it * 64 * PSTR) = rh[k];
        }
    };

    // ---- per-lane fragment bases
    int p0[FM][3], bofs[FN];     // pixel fragment base per tap row: everything else of a read address is an immediate
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wave * 64 + i * 32 + frow;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
            p0[i][ty] = OFF_H + (((ml >> twl) + ty) * HC + ((ml + lane_rot(ml >> twl, twl)) & (TW - 1))) * PSTR + fh * 16;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) bofs[j] = OFF_B + (j * 32 + frow) * PSTR + fh * 16;

    bf16x8 fa[2][FM], fb[2][FN];
    auto read_frags = [&](auto setc, auto tyc, int hofs, int bofs_imm) {
        constexpr int set = decltype(setc)::value, ty = decltype(tyc)::value;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(smem + p0[i][ty] + hofs);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(smem + bofs[j] + bofs_imm);
    };
    f32x16 acc[FM][FN];
    auto mma = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
#if defined(DS_HALO2_ABL_16x16)
        // timing experiment only (wrong results): the same FLOPs issued as 16x16x32 MFMAs on the same registers
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                f32x4 c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], c1, 0, 0, 0);
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
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: ONE memory round trip.  The small operands of the GroupNorm fold go first (statistics partials, the
    // fold-table entries this thread will combine), then the halo and the first weight tiles; the float64 reduction of the
    // partials and the shift table are computed while the big loads are still in flight.
    const bool fold = p.gn_ab != nullptr || p.gn_part != nullptr;
    const int ncls = fold ? p.ncls : 1;
    constexpr int ST_IT = (9 * BN + NT - 1) / NT;      // 4 shift-table entries per thread at most
    float t1v[ST_IT], t2v[ST_IT];
    if (ksplit == 1) {
#pragma unroll
        for (int k = 0; k < ST_IT; ++k) {
            const int e = tid + k * NT, cls = e / BN, n = n0 + e - cls * BN;
            t1v[k] = 0.f;
            t2v[k] = 0.f;
            if (e < ncls * BN && n < p.Cout) {
                if (fold) {
                    t1v[k] = DS_LD(float, p.fold_t1 + cls * p.Cout + n, DS_BX_T1);
                    t2v[k] = DS_LD(float, p.fold_t2 + cls * p.Cout + n, DS_BX_T2);
                } else if (p.bias) t1v[k] = DS_LD(float, p.bias + n, DS_BX_BIAS);
                if (NR > 0 && p.res_bias) t1v[k] += DS_LD(float, p.res_bias + n, DS_BX_AUX2);
            }
        }
    }
    const long st_p1 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;    // setup + small loads issued
    u32x4 rh2[HH];
    h_so = NR > 0 ? 0u : (unsigned)cc_lo * 64u;
    load_halo_to(rh2, I0{});
    load_halo(I1{});
    load_b(I0{});
    load_b(I1{});
    float gn_a = 1.f, gn_am = 0.f;
    if (ksplit == 1) {
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, gn_a, gn_am);
        else if (p.gn_ab) {
            gn_a = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
            gn_am = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
        }
#pragma unroll
        for (int k = 0; k < ST_IT; ++k) {
            const int e = tid + k * NT;
            if (e < ncls * BN) shl[e] = t1v[k] - gn_am * t2v[k];
        }
    }
    const long st_p2 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;    // statistics reduced, shift table written
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
        char* const h = smem + OFF_H;
#pragma unroll
        for (int k = 0; k < HH; ++k) *reinterpret_cast<u32x4*>(h + lds0 + k * 64 * PSTR) = rh2[k];
    }
    store_halo(I0{}, I1{});
    store_b(I0{}, I0{});
    store_b(I1{}, I1{});
    load_b(I2{});
    load_b(I0{});
    load_b(I1{});
    __syncthreads();

    // ---- fused res_conv: NR one-step chunks at the centre tap (NR % 3 == 0, so the weight ring is back at phase 0 when the
    // nine-tap chunks start; the 3x3 tiles follow the res tiles in wpk).  A one-step chunk cannot prefetch the next step's
    // fragments across the barrier (that halo buffer is being written during the step), so each step pays one LDS read
    // latency: NR is 3 .. 24 against 27 .. 216 nine-tap steps.  The last res step stages the first 3x3 chunk's halo.
    if (NR > 0) {
        set_res_src(1);
        load_halo_to(rh, I0{});
        load_halo_to(rh + HH, I1{});
        auto res_step = [&](auto hbc, auto phc, int r) {
            constexpr int hb = decltype(hbc)::value, ph = decltype(phc)::value, rs = (ph + 2) % 3;
            // (order pinned with sched_barrier: left alone, the scheduler hoists the six refill loads above the six stores of
            // the same staging registers, doubles their live range and spills into the nine-tap loop)
            read_frags(I0{}, I1{}, hb * HALO_BYTES + PSTR, ph * B_BYTES);
            read_frags(I1{}, I1{}, hb * HALO_BYTES + PSTR + 32, ph * B_BYTES + 32);
            __builtin_amdgcn_sched_barrier(0);
            {                                          // stage chunk r + 1 (requested one step ago) into the other halo buffer
                char* const h = smem + OFF_H + (hb ^ 1) * HALO_BYTES;
#pragma unroll
                for (int it = 0; it < H_IT; ++it) {
                    if (it == H_IT - 1) *reinterpret_cast<u32x4*>(h + hst_last) = rh[it];
                    else *reinterpret_cast<u32x4*>(h + lds0 + it * 64 * PSTR) = rh[it];
                }
            }
            store_b(std::integral_constant<int, rs>{}, std::integral_constant<int, rs>{});
            __builtin_amdgcn_sched_barrier(0);
            load_b(std::integral_constant<int, rs>{});
            if (r + 2 <= NR) {                         // r + 2 == NR: the first 3x3 chunk (full halo of the 3x3 input)
                set_res_src(r + 2);
                __builtin_amdgcn_sched_barrier(0);
                load_halo_to(rh, I0{});
                load_halo_to(rh + HH, I1{});
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(I0{});
            mma(I1{});
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        for (int r = 0; r < NR; r += 6) {
            res_step(I0{}, I0{}, r);
            res_step(I1{}, I1{}, r + 1);
            res_step(I0{}, I2{}, r + 2);
            if (r + 3 < NR) {
                res_step(I1{}, I0{}, r + 3);
                res_step(I0{}, I1{}, r + 4);
                res_step(I1{}, I2{}, r + 5);
            }
        }
        // acc_res + a * acc_3x3 = a * (acc_res / a + acc_3x3): the epilogue applies the GroupNorm factor a to the whole sum
        const float inv_a = 1.0f / gn_a;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv_a;
    }
    const int par = NR & 1;                // halo buffer of the first 3x3 chunk
    if (par) read_frags(I0{}, I0{}, HALO_BYTES, 0);
    else read_frags(I0{}, I0{}, 0, 0);

    long st_lgkm = 0, st_bar = 0;
    // ---- main loop: chunks x 9 taps.  hbuf (halo double buffer) and every ring index are compile-time constants.
    auto chunk = [&](auto hbufc, int cc) {
        constexpr int hbuf = decltype(hbufc)::value;
        h_so = (unsigned)(cc_lo + (cc + 1 < NCC ? cc + 1 : cc)) * 64u;      // chunk prefetched during this one (a dummy re-read at the end)
        auto step = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int rs = (tap + 2) % 3;                  // ring slot stored this step (tile s + 2), then refilled with tile s + 5
            constexpr int ty = tap / 3, tx = tap % 3;
            constexpr int nty = (tap + 1) / 3, ntx = (tap + 1) % 3;
            // region 1: LDS writes of this step, loads for later steps, fragment reads of k-substep 1, MFMAs of k-substep 0
            if constexpr (tap == 3) store_halo(std::integral_constant<int, hbuf ^ 1>{}, I0{});
            if constexpr (tap == 7) store_halo(std::integral_constant<int, hbuf ^ 1>{}, I1{});
            store_b(std::integral_constant<int, rs>{}, std::integral_constant<int, rs>{});
            load_b(std::integral_constant<int, rs>{});
            if constexpr (tap == 1) load_halo(I0{});
            if constexpr (tap == 4) load_halo(I1{});
            read_frags(I1{}, std::integral_constant<int, ty>{}, hbuf * HALO_BYTES + tx * PSTR + 32, (tap % 3) * B_BYTES + 32);
            mma(I0{});
            // one or two LDS / VMEM instructions per MFMA gap: the fragment reads lead (the next cluster needs them), the
            // LDS writes (only this step's barrier waits for them) and the loads for later steps follow
            constexpr int NW1 = 2 + ((tap == 3 || tap == 7) ? HH : 0), NV1 = 2 + ((tap == 1 || tap == 4) ? HH : 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSW, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSW, NW1 > 2 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            if constexpr (NW1 > 2) __builtin_amdgcn_sched_group_barrier(SG_DSW, 2, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, NV1 > 2 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            if constexpr (NV1 > 2) __builtin_amdgcn_sched_group_barrier(SG_VMEM, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            // region 2: fragment reads of the next step's k-substep 0, MFMAs of k-substep 1
            if constexpr (tap < 8) read_frags(I0{}, std::integral_constant<int, nty>{}, hbuf * HALO_BYTES + ntx * PSTR, ((tap + 1) % 3) * B_BYTES);
            else read_frags(I0{}, I0{}, (hbuf ^ 1) * HALO_BYTES, 0);
            mma(I1{});
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            // this step's LDS writes precede (in program order, hence in completion order) the five reads just issued:
            // waiting until at most five LDS operations are outstanding retires the writes and leaves the reads in flight
            // (no per-step stamps here: s_memtime shares lgkmcnt with the LDS reads and would drain them every step)
            asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
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
    };
    long st_t0 = 0, st_r0 = 0;
    if constexpr (DS_STAMP) {
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) alone: the stamps must not leave the loop's first LDS waits at lgkmcnt(0)
    }
    // (an if / else between the two instantiations inside the loop made the register allocator keep two copies of the
    // accumulators at the join and spill; an odd start is peeled instead)
    int cc0 = 0;
    if (par) {
        chunk(I1{}, 0);
        cc0 = 1;
    }
    for (int cc = cc0; cc < NCC; cc += 2) {
        chunk(I0{}, cc);
        if (cc + 1 < NCC) chunk(I1{}, cc + 1);
    }
    if constexpr (DS_STAMP) {
        const long st_t1 = __builtin_amdgcn_s_memtime();
        if (p.slab && ksplit == 1 && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
            d[0] = st_t1 - st_t0; d[1] = st_p1 - st_k0; d[2] = st_p2 - st_k0; d[3] = nsteps;
            (void)st_lgkm; (void)st_bar;
            d[4] = st_r0 - st_k0; d[5] = __builtin_amdgcn_s_memrealtime() - st_r0; d[6] = st_k0;
        }
    }

    // ---- epilogue
    auto coord = [&](int ml) {
        ConvCoord c;
        c.ho = h0 + (ml >> twl);
        c.wo = w0 + ((ml + lane_rot(ml >> twl, twl)) & (TW - 1));
        c.ok = c.ho < p.H && c.wo < p.W;
        c.pix = c.ho * p.W + c.wo;
        return c;
    };
    float s1 = 0.f, s2 = 0.f;
    if (ksplit > 1) {
        // raw fp32 partial sums of this K slice -> slab[kz][b]; bias / fold / activation / residual / statistics happen in
        // ds_conv_splitk_reduce
        ds_conv_params q = p;
        q.out = p.slab;
        q.out_C = (p.Cout + 7) / 8 * 8;
        q.out_c0 = 0;
        conv_epilogue_t_body<float, FM, FN, BN, DS_ACT_NONE, false, true>(q, acc, kz * p.B + b, n0, 0, wave * 64, p.H * p.W, shl, coord, s1, s2, 1.f);
        return;
    }
    const long st_e1 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    conv_epilogue_t<bf16, FM, FN, BN>(p, acc, b, n0, 0, wave * 64, p.H * p.W, shl, coord, s1, s2, gn_a);
    const long st_e2 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    __syncthreads();
    if (p.stats_part) {
        const int parts = gridDim.x * gridDim.y;
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2);
    }
    if constexpr (DS_STAMP) {
        if (p.slab && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
            d[7] = __builtin_amdgcn_s_memrealtime() - st_k0;
            (void)st_e1; (void)st_e2;
        }
    }
}

int halo2_twl(int W) {
    int twl = 3;
    while ((1 << twl) < W && twl < 5) ++twl;
    return twl;
}

}  // namespace

int ds_conv3x3_halo2_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv3x3_halo2: bf16 only");
    DS_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad_h == 1 && p->pad_w == 1 && !p->transposed,
               "conv3x3_halo2: 3x3 stride 1 pad 1 only");
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0, "conv3x3_halo2: single source, Cin multiple of 32 (got %d+%d)", p->C0, p->C1);
    DS_REQUIRE(p->Ho == p->H && p->Wo == p->W && !p->out_nchw_f32, "conv3x3_halo2: same-size NHWC output only");
    DS_REQUIRE(p->cout_pad % BN == 0 && p->wk_order == 1, "conv3x3_halo2: cout_pad %% 96 == 0 and chunk-major weights (wk_order = 1)");
    DS_REQUIRE(p->ksplit <= 1 || (p->slab && (p->C0 / 32) % p->ksplit == 0),
               "conv3x3_halo2: ksplit=%d needs a slab and must divide the %d channel chunks", p->ksplit, p->C0 / 32);
    if (p->res_steps) {
        DS_REQUIRE(p->ksplit <= 1 && !p->res, "conv3x3_halo2: a fused res_conv excludes split-K and a residual tensor");
        DS_REQUIRE(p->res_src0 && p->res_C0 > 0 && p->res_C0 % 32 == 0 && p->res_C1 % 32 == 0 && p->res_steps == (p->res_C0 + p->res_C1) / 32 &&
                       p->res_steps % 3 == 0,
                   "conv3x3_halo2: res_conv channels (%d,%d) must be multiples of 32 (96 in total) and res_steps = their chunks", p->res_C0, p->res_C1);
        DS_REQUIRE(p->res_C1 == 0 || (p->res_src1 && p->res_H1 > 0 && p->res_W1 > 0), "conv3x3_halo2: second res_conv source incomplete");
        DS_REQUIRE(ds_aligned16(p->res_src0) && (!p->res_C1 || ds_aligned16(p->res_src1)), "conv3x3_halo2: res_conv sources must be 16-byte aligned");
    }
    DS_REQUIRE((long long)p->H * p->W * (p->C0 > p->res_C0 ? p->C0 : p->res_C0) * 2 < (1ll << 31) &&
                   (long long)((p->C0 / 32) * 9 + p->res_steps) * p->cout_pad * 64 < (1ll << 31),
               "conv3x3_halo2: one sample / the packed weights must stay below 2 GiB (32-bit buffer offsets)");
    DS_SET_MAX_LDS(conv3x3_halo2_kernel, LDS_BYTES, "conv3x3_halo2");
    const int twl = halo2_twl(p->W), TW = 1 << twl, TH = BM >> twl;
    dim3 grid(((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW), p->cout_pad / BN, p->B * (p->ksplit > 1 ? p->ksplit : 1));
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_HALO);
        ds_conv_bounds_table(*p, DS_K_CONV_HALO, grid.x * grid.y, &h.t);
        h.set(DS_BX_W, p->wpk, (long long)((p->C0 / 32) * 9 + p->res_steps) * p->cout_pad * 64);
        h.set(DS_BX_AUX0, p->res_steps ? p->res_src0 : nullptr, (long long)p->B * p->H * p->W * p->res_C0 * 2);
        h.set(DS_BX_AUX1, p->res_C1 ? p->res_src1 : nullptr, (long long)p->B * p->res_H1 * p->res_W1 * p->res_C1 * 2);
        h.set(DS_BX_AUX2, p->res_bias, (long long)p->Cout * 4);
        h.publish(st);
    }
#endif
    hipLaunchKernelGGL(conv3x3_halo2_kernel, grid, dim3(NT), LDS_BYTES, st, *p, twl, 65536 / (TW + 2) + 1);
    DS_CHECK_LAUNCH("conv3x3_halo2");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_halo2(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
