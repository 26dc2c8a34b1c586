// 4x4 stride-2 convolutions of the U-Net (Downsample: Conv2d(dim, dim, 4, 2, 1); Upsample: ConvTranspose2d(dim, dim, 4, 2, 1),
// diffusion_components.py:88-93) on the conv3x3_halo3 pipeline: bf16, 16x16x32 MFMAs from an XOR-swizzled LDS halo (gfx950).
//
// Both are "quad" convolutions over a base grid of pixels (i, j):
//   transposed: the grid is the INPUT image; output pixel (2i + py, 2j + px) — phase (py, px) — is a 2x2 convolution over input
//               pixels (i + py + a - 1, j + px + b - 1), a, b in {0, 1}, with kernel element (3 - py - 2a, 3 - px - 2b);
//               the four phases are four groups of output channels of one launch (N = 4 * Cout, an N-block = one phase);
//   strided:    the grid is the OUTPUT image; the input splits into four parity planes S_pq(r, c) = in(2r + p, 2c + q); plane
//               (p, q) contributes a 2x2 convolution over S_pq(i - p + a, j - q + b) with kernel element (1 - p + 2a, 1 - q + 2b):
//               the four planes are four groups of K chunks of one launch.
// In halo coordinates (halo row 0 = grid row i0 - 1) every step reads a tap of the same 3x3 neighbourhood the 3x3 kernel uses,
// but only FOUR taps per 32-channel chunk: window origin (py, px) resp. (1 - p, 1 - q).  The generic implicit-GEMM kernel
// gathered these operands per tap through registers and ran the six layers at 525-690 TFLOP/s (8.5 % of a step).
//
// Pipeline = conv3x3_halo3.hip's with a chunk of 4 steps: weight tiles [chunk][tap][cout_pad][32] (wk_order = 2, packed by
// diffusynth_amd/engine.py:pack_quad_weights) through the 3-slot register ring into the 3-buffer LDS ring (4 = 1 mod 3: the ring
// phase advances by one per chunk, so the loop is unrolled over 6 chunks = ring phase x halo buffer; the chunk count must be a
// multiple of 6); the next chunk's halo is staged in two halves (loaded at taps 3 and 0, stored at taps 1 and 2: its first read
// is the fragment prefetch issued in tap 3).  The window origin is constant over a block: a transposed block has one phase, and
// the strided kernel stores plane (p, q)'s halo shifted by (+p, +q) pixels in LDS, so that its window always starts at (1, 1) —
// the four tap addresses are per-lane registers computed once, only the halo STORE base (one register) changes with the plane.
#include "common.hpp"
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif

#include "conv_halo3_common.hpp"

#ifndef DS_QUAD_ROWS
#define DS_QUAD_ROWS 1          // fp32 output mode: line-sized stores through an LDS tile (a lane's fp32 run is 32 bytes, two instructions)
#endif
#ifndef DS_QUAD_ROWS_BF16
#define DS_QUAD_ROWS_BF16 0     // bf16: 64-byte runs straight from the registers (EPI_CH map) measure the same as the staged form
#endif

namespace {

constexpr int QHALO_BYTES = HALO_BYTES + 1024;            // + 16 pixels: the shifted store of a strided plane (up to one row + one pixel)
constexpr int QLDS_BYTES = OFF_H + 2 * QHALO_BYTES;        // 81856 <= 81920: two blocks per CU
static_assert(QLDS_BYTES <= 81920, "two blocks per CU");
static_assert(4 * EPI_F32_WAVE <= 2 * QHALO_BYTES, "the epilogue's staging tiles fit the (dead) halo buffers");

template <int TWL>
__global__ __launch_bounds__(NT, DS_MINBLK) void conv_quad_halo3_kernel(const ds_conv_params p) {
    using G = HG<TWL>;
    constexpr int TW = G::TW, TH = G::TH, HCP = G::HCP, NPX = G::NPX, H_IT = G::H_IT, HH0 = G::HH0, HH1 = G::HH1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const shl = reinterpret_cast<float*>(smem + OFF_SHL);
    float* const red = reinterpret_cast<float*>(smem + OFF_B);      // reused after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);       // (wave-uniform: the epilogue's copy lives in an SGPR)
    const int m = lane & 15, q = lane >> 4;
    const bool tr = p.transposed != 0;
    const int Hg = p.Ho, Wg = p.Wo;                                 // base grid (transposed: = input image, strided: = output image)
    const int tiles_w = (Wg + TW - 1) >> TWL;
    // XCD-chunked block order (see conv3x3_halo3.hip)
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    int wid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
    const int by = wid % gy, bxz = wid / gy, bx = bxz % gx, bz = bxz / gx;
    const int th = bx / tiles_w, tw = bx - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    // split-K (r04: the 32 x 8 / 64 x 16-level Down / Upsample of a small batch were one to eight blocks of up to 216 serial chunks — 138 us per
    // launch at batch 1): blockIdx.z = K slice * B + sample; a slice runs a contiguous range of the chunk sequence and stores raw fp32
    // partial sums to p.slab, ds_conv_splitk_reduce adds the slices and runs the epilogue (conv3x3_halo3.hip)
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int kz = bz / p.B, b = bz - kz * p.B, n0 = by * BN;        // (slice-major: bz = kz * B + b is the slab's batch index as it stands)
    // split-precision input (flags & DS_CONV_F_SPLIT_IN, see conv3x3_halo3.hip): the source holds 2C bf16 channels (hi plane, lo plane);
    // a plane's chunk sequence is hi, lo, hi against [W_hi | W_hi | W_lo]
    const bool split_in = (p.flags & DS_CONV_F_SPLIT_IN) != 0;
    const int Cin = p.C0, NSRC = Cin >> 5;          // channels per source pixel; 32-channel chunks the source holds
    const int CC = split_in ? NSRC + (NSRC >> 1) : NSRC;   // chunks per parity plane of the K loop
    const int NCC = (tr ? CC : 4 * CC) / ksplit;    // chunks of this block's K loop
    const int k0 = kz * NCC;                        // first chunk of the slice in the launch's chunk sequence (plane-major)
    const int par0 = tr ? 0 : k0 / CC, c0 = k0 - par0 * CC;
    const int phase = tr ? n0 / p.Cout : 0;         // transposed: output phase of this N-block

    const char* const wbase = reinterpret_cast<const char*>(p.wpk);
    const unsigned wbytes = (unsigned)(NCC * ksplit * 4) * p.cout_pad * 64;
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), (short)0, (int)wbytes, 0x00020000);

    // ---- input halo: (TH + 2) x (TW + 2) grid pixels around the tile, 32 channels of one chunk (one parity plane)
    const char* const hbase = reinterpret_cast<const char*>(p.src0) + (size_t)b * p.H * p.W * Cin * 2;
    const rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(hbase), (short)0, (int)((unsigned)p.H * p.W * Cin * 2), 0x00020000);
    unsigned hvo[H_IT];
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
        const int slot = tid + it * NT, hp = slot >> 2, dq = slot & 3;
        const int hr = hp / HCP, hc = hp - hr * HCP;
        // (arithmetic, not nested ifs — see conv3x3_halo3.hip; an offset with bit 31 set = VOFF_NONE)
        const int r = h0 + hr - 1, c = w0 + hc - 1;
        // transposed: only the (TH + 1) x (TW + 1) window of this phase is ever read
        const unsigned outside = (unsigned)(hr < (phase >> 1)) | (unsigned)(hr > (phase >> 1) + TH) | (unsigned)(hc < (phase & 1)) | (unsigned)(hc > (phase & 1) + TW);
        const unsigned bad = (unsigned)(hp >= NPX) | (unsigned)(hc >= TW + 2) | ((unsigned)tr & outside) |
                             (unsigned)((unsigned)r >= (unsigned)Hg) | (unsigned)((unsigned)c >= (unsigned)Wg);
        const unsigned off = tr ? (unsigned)((r * p.W + c) * Cin + dq * 8) * 2u : (unsigned)((2 * r * p.W + 2 * c) * Cin + dq * 8) * 2u;
        hvo[it] = (off & 0x7fffffffu) | (bad << 31);
    }
    // scalar byte offset of chunk (plane par, 32-channel group c32)
    auto chunk_so = [&](int par, int c32) -> unsigned {
        const int sc = c32 < NSRC ? c32 : c32 - NSRC;         // third group of a split input: the hi chunks again
        return tr ? (unsigned)sc * 64u : (unsigned)(((par >> 1) * p.W + (par & 1)) * Cin + sc * 32) * 2u;
    };

    // halo store base of plane par: slot pixel hp = tid >> 2 (+ 64 per iteration: bit2 unchanged) shifted by (p, q) pixels
    auto halo_store_base = [&](int par) {
        const int hp = (tid >> 2) + (tr ? 0 : (par >> 1) * HCP + (par & 1));
        return OFF_H + hp * PSTR + (((tid & 3) ^ (((hp >> 2) & 1) << 1)) << 4);
    };
    int lds_h = halo_store_base(par0);
    // Weight tile = 384 pieces of 16 bytes for 256 threads.  Second piece of a thread: piece tid + 256 for waves 0-1 and tid + 128 for waves
    // 2-3 (these REPEAT pieces 256..383: the same bytes to the same place) — a wave-uniform distance from the first piece in global memory
    // and in LDS (rows + 64 / + 32: W_SWZ is unchanged by multiples of 32 rows), so the second load / store needs no offset registers of
    // its own (scalar offset resp. one v_add per step; two registers less in a kernel at its 256-register budget).
    const int wr0 = tid >> 2;
    const int bst0 = OFF_B + wr0 * PSTR + (((tid & 3) ^ W_SWZ(wr0)) << 4);
    const unsigned wvo0 = (unsigned)tid * 16u;
    const int w2rows = wave_s < 2 ? 64 : 32;                 // SGPR: distance of the second piece in rows (= 64 bytes each, both sides)
    static_assert(W_SWZ(5) == W_SWZ(5 + 32) && W_SWZ(13) == W_SWZ(13 + 64) && BN * 4 == NT + 128, "second-piece distance keeps the swizzle");
    const unsigned wstep = (unsigned)p.cout_pad * 64u;
    // (the prefetches of a slice's last steps run on into the next slice's chunks — valid memory, never used; only the end of the whole
    // sequence is clamped, as without split-K)
    const unsigned w_first = (unsigned)n0 * 64u + (unsigned)(k0 * 4) * wstep, w_last = (unsigned)n0 * 64u + (unsigned)(NCC * ksplit * 4 - 1) * wstep;
    unsigned w_pf = w_first;

    u32x4 rb[2][2], rhA[HH0], rhB[HH1];     // weight tiles: TWO register slots (tile s + 2 is stored at step s, its slot refilled with tile s + 4)
    auto load_b = [&](auto slotc) {
        constexpr int sl = decltype(slotc)::value;
        rb[sl][0] = buf_ld16(rs_w, wbase, wvo0, w_pf, DS_BX_W);
        rb[sl][1] = buf_ld16(rs_w, wbase, wvo0, w_pf + (unsigned)w2rows * 64u, DS_BX_W);
        const unsigned nx = w_pf + wstep;
        w_pf = nx < w_last ? nx : w_last;
    };
    auto store_b = [&](auto slotc, auto bufc) {
        constexpr int sl = decltype(slotc)::value, buf = decltype(bufc)::value;
        *reinterpret_cast<u32x4*>(smem + buf * B_STRIDE + bst0) = rb[sl][0];
        int d2 = w2rows * PSTR;
        asm volatile("" : "+s"(d2));             // (opaque: hoisted out of the K loop, bst0 + d2 is a register again)
        *reinterpret_cast<u32x4*>(smem + buf * B_STRIDE + bst0 + d2) = rb[sl][1];
    };
    auto load_half = [&](u32x4* dst, auto halfc, unsigned so) {
        constexpr int half = decltype(halfc)::value, n = half ? HH1 : HH0;
#pragma unroll
        for (int k = 0; k < n; ++k) dst[k] = buf_ld16(rs_h, hbase, hvo[half * HH0 + k], so, DS_BX_SRC0);
    };
    auto store_half = [&](const u32x4* src, auto bufc, auto halfc) {
        constexpr int buf = decltype(bufc)::value, half = decltype(halfc)::value, n = half ? HH1 : HH0;
#pragma unroll
        for (int k = 0; k < n; ++k) *reinterpret_cast<u32x4*>(smem + lds_h + buf * QHALO_BYTES + (half * HH0 + k) * 64 * PSTR) = src[k];
    };

    // ---- per-lane fragment addresses
    int hp0[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
        else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }
        hp0[i] = row_l * HCP + col_l;
    }
    // Fragment addresses (halo buffer 0) of the four taps of window origin (oy, ox) = (py, px) resp. (1, 1): ONE base per pixel tile (tap 0)
    // and, as in conv3x3_halo3, a per-lane XOR mask for the swizzle bit (address bit 5 = bit2 of the halo pixel): a step down the window
    // adds HCP = 4 x odd pixels (bit2 always flips), a step right flips it where the pixel's column is 3 mod 4 — the column of every tile is
    // m + ox modulo 4.  tap (ty, tx) = (base ^ mask(ty, tx)) + (ty * HCP + tx) * PSTR.  (Sixteen address registers — one per tap and
    // tile — put this kernel 13 registers over its budget: 52-68 bytes of scratch per lane since round 2.)
    static_assert((HCP & 3) == 0 && ((HCP >> 2) & 1) == 1, "tap-row swizzle flip assumes HCP / 4 odd");
    const int oy0 = tr ? (phase >> 1) : 1, ox0 = tr ? (phase & 1) : 1;
    int xa0[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        const int hp = hp0[i] + oy0 * HCP + ox0;
        xa0[i] = OFF_H + hp * PSTR + ((q ^ (((hp >> 2) & 1) << 1)) << 4);
    }
    const int xm1 = (((m + ox0) & 3) == 3) << 5;
    auto xaddr = [&](auto tapc, int i) {
        constexpr int t = decltype(tapc)::value;
        int a = xa0[i];
        // (opaque to loop-invariant code motion: hoisted, the twelve XORed addresses are sixteen live registers again)
        if constexpr (t != 0) asm volatile("" : "+v"(a));
        if constexpr (t == 1) a ^= xm1;
        if constexpr (t == 2) a ^= 32;
        if constexpr (t == 3) a ^= xm1 ^ 32;
        return a + ((t >> 1) * HCP + (t & 1)) * PSTR;
    };
    const int bw = OFF_B + W_ROW0(m) * PSTR + ((q ^ ((-(m >> 2)) & 3)) << 4);      // + EPI_CH(j) rows for tile j

    bf16x8 fx[2][XT], fw[WT];
    auto read_w = [&](int j, int imm) { fw[j] = *reinterpret_cast<const bf16x8*>(smem + bw + EPI_CH(j) * PSTR + imm); };
    f32x4 acc[XT][WT];
    auto mma_j = [&](auto setc, int j) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < XT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[set][i], acc[i][j], 0, 0, 0);   // D^T = W . X^T
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: bias row of the shift table, halo of chunk 0 (both halves) and half 0 of chunk 1, weight tiles 0..4
    constexpr int ST_IT = (10 * BN + NT - 1) / NT;
    float t1v[ST_IT];
#pragma unroll
    for (int k = 0; k < ST_IT; ++k) {
        const int e = tid + k * NT, n = n0 + e - phase * p.Cout;      // e < BN: row 0 = bias of this block's channels
        t1v[k] = 0.f;
        if (e < BN && n < p.Cout && p.bias && ksplit == 1) t1v[k] = DS_LD(float, p.bias + n, DS_BX_BIAS);      // (a K slice: the reduce adds the bias)
    }
    // prefetch pointers: chunk (par, c32) whose half 0 / half 1 is loaded next (clamped at the last chunk: dummy re-reads)
    int parA = par0, cA = c0, parB = par0, cB = c0;
    auto advance = [&](int& par, int& c32) {
        const bool last = par == (tr ? 0 : 3) && c32 == CC - 1;
        if (!last) {
            ++c32;
            if (c32 == CC) { c32 = 0; ++par; }
        }
    };
    u32x4 rh2[HH0];
    load_half(rh2, I0{}, chunk_so(par0, c0));
    load_half(rhB, I1{}, chunk_so(par0, c0));
    load_b(I0{});
    load_b(I1{});
    advance(parA, cA);                       // -> chunk 1
    load_half(rhA, I0{}, chunk_so(parA, cA));
    advance(parA, cA);                       // -> chunk 2: loaded in tap 3 of chunk 0
    advance(parB, cB);                       // -> chunk 1: loaded in tap 0 of chunk 0
#pragma unroll
    for (int k = 0; k < ST_IT; ++k) {
        const int e = tid + k * NT;
        if (e < 10 * BN) shl[e] = t1v[k];
    }
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    store_half(rh2, I0{}, I0{});
    store_half(rhB, I0{}, I1{});
    lds_h = halo_store_base(parB);           // chunk 1 is staged during chunk 0
    store_b(I0{}, I0{});
    store_b(I1{}, I1{});
    load_b(I0{});
    load_b(I1{});
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XT; ++i) fx[0][i] = *reinterpret_cast<const bf16x8*>(smem + xaddr(I0{}, i));
#pragma unroll
    for (int j = 0; j < WT; ++j) read_w(j, 0);

    // ---- main loop: chunks x 4 taps; hbuf (halo double buffer) and ph (weight ring phase = chunk mod 3) are compile-time
    auto chunk = [&](auto hbufc, auto phc) {
        constexpr int hbuf = decltype(hbufc)::value, ph = decltype(phc)::value;
        const unsigned soA = chunk_so(parA, cA), soB = chunk_so(parB, cB);
        auto step = [&](auto tapc) {
            constexpr int t = decltype(tapc)::value;
            constexpr int rs = (ph + t + 2) % 3, rr = t & 1;   // LDS slot of tile s + 2 (stored this step); its register slot, refilled with tile s + 4
            constexpr int cur = t & 1;
            constexpr int nslot = (ph + t + 1) % 3, nhb = t == 3 ? (hbuf ^ 1) : hbuf;
            constexpr int nW = 2 + (t == 1 ? HH0 : (t == 2 ? HH1 : 0)), nV = 2 + (t == 0 ? HH1 : (t == 3 ? HH0 : 0));
            if constexpr (t == 1) store_half(rhA, std::integral_constant<int, hbuf ^ 1>{}, I0{});
            if constexpr (t == 2) store_half(rhB, std::integral_constant<int, hbuf ^ 1>{}, I1{});
            store_b(std::integral_constant<int, rr>{}, std::integral_constant<int, rs>{});
            load_b(std::integral_constant<int, rr>{});
            if constexpr (t == 0) load_half(rhB, I1{}, soB);
            if constexpr (t == 3) load_half(rhA, I0{}, soA);
            mma_j(std::integral_constant<int, cur>{}, 0);
            read_w(0, nslot * B_STRIDE);
#pragma unroll
            for (int i = 0; i < XT; ++i)
                fx[cur ^ 1][i] = *reinterpret_cast<const bf16x8*>(smem + xaddr(std::integral_constant<int, (t + 1) & 3>{}, i) + nhb * QHALO_BYTES);
#pragma unroll
            for (int j = 1; j < WT; ++j) {
                mma_j(std::integral_constant<int, cur>{}, j);
                read_w(j, nslot * B_STRIDE);
            }
#pragma unroll
            for (int k = 0; k < nW; ++k) { SGB(SG_MFMA, 1); SGB(SG_DSW, 1); }
#pragma unroll
            for (int k = 0; k < nV; ++k) { SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); }
            constexpr int L0 = nW + nV;
#pragma unroll
            for (int k = 0; k < 5; ++k) { SGB(SG_MFMA, 1); SGB(SG_DSR, 1); }
            constexpr int L1 = L0 + 5;
#pragma unroll
            for (int j = 1; j < WT; ++j) {
                const int before = (4 * j + 4 > L1 + j - 1 ? 4 * j + 4 : L1 + j - 1);
                const int prev = j == 1 ? L1 : (4 * (j - 1) + 4 > L1 + j - 2 ? 4 * (j - 1) + 4 : L1 + j - 2);
                if (before - prev == 1) SGB(SG_MFMA, 1);
                else if (before - prev == 2) SGB(SG_MFMA, 2);
                else if (before - prev == 3) SGB(SG_MFMA, 3);
                else if (before - prev == 4) SGB(SG_MFMA, 4);
                else if (before - prev == 5) SGB(SG_MFMA, 5);
                else if (before - prev == 6) SGB(SG_MFMA, 6);
                else if (before - prev == 7) SGB(SG_MFMA, 7);
                SGB(SG_DSR, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#if DS_BOUNDS || defined(DS_LGKM0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the checker's extra code may reorder the step: no counted wait in this build
#else
            asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
#endif
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        // between chunks (uniform scalar work): advance the prefetch pointers; a new parity plane moves the halo store base
        advance(parA, cA);
        const int par_old = parB;
        advance(parB, cB);
        if (parB != par_old) lds_h = halo_store_base(parB);
    };
    for (int cc = 0; cc < NCC; cc += 6) {
        chunk(I0{}, I0{});
        chunk(I1{}, I1{});
        chunk(I0{}, I2{});
        chunk(I1{}, I0{});
        chunk(I0{}, I1{});
        chunk(I1{}, I2{});
    }

    // ---- epilogue: bias, bf16 store.  Transposed: pixel (2i + py, 2j + px) of the 2H x 2W image, channel n - phase * Cout.
    // The thread's coordinates are derived AGAIN — lane from mbcnt, the wave index from an SGPR: reusing the prologue's tid / lane / wave /
    // m kept nine registers alive across the K loop, in scratch (the kernel is at its 256-register budget; zero scratch is pinned by
    // tests/test_isa_schedule_cpu.py).
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wave_e = wave_s, m_e = lane_e & 15;
    auto coord = [&](int i) {
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave_e + (i >> 1); col_l = 16 * (i & 1) + m_e; }
        else if constexpr (TWL == 4) { row_l = 4 * wave_e + i; col_l = m_e; }
        else { row_l = 8 * wave_e + i + 4 * (m_e >> 3); col_l = m_e & 7; }
        ConvCoord c;
        c.ho = h0 + row_l;
        c.wo = w0 + col_l;
        c.ok = c.ho < Hg && c.wo < Wg;
        c.pix = tr ? (2 * c.ho + (phase >> 1)) * (2 * Wg) + 2 * c.wo + (phase & 1) : c.ho * Wg + c.wo;
        return c;
    };
    auto coord2 = [&](int i, int mm) {                                  // pixel (tile i, lane mm): the contiguous side of halo3_epilogue_rows
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave_e + (i >> 1); col_l = 16 * (i & 1) + mm; }
        else if constexpr (TWL == 4) { row_l = 4 * wave_e + i; col_l = mm; }
        else { row_l = 8 * wave_e + i + 4 * (mm >> 3); col_l = mm & 7; }
        ConvCoord c;
        c.ho = h0 + row_l;
        c.wo = w0 + col_l;
        c.ok = c.ho < Hg && c.wo < Wg;
        c.pix = tr ? (2 * c.ho + (phase >> 1)) * (2 * Wg) + 2 * c.wo + (phase & 1) : c.ho * Wg + c.wo;
        return c;
    };
    ds_conv_params qp = p;
    qp.gn_ab = nullptr;
    qp.gn_part = nullptr;
    qp.res = nullptr;
    if (tr) {
        qp.out_c0 = p.out_c0 - phase * p.Cout;
        qp.Cout = p.cout_pad;
    }
    float s1 = 0.f, s2 = 0.f;
    const int outHW = tr ? 4 * Hg * Wg : Hg * Wg;
    const bool raw = p.ksplit > 1;
    if (raw) {                               // fp32 partial sums of this K slice -> slab[kz][b][pixel][roundup(Cout, 8)] through the fp32 epilogue
        qp.out = p.slab;
        qp.out_C = (p.Cout + 7) / 8 * 8;
        qp.out_c0 = tr ? -phase * p.Cout : 0;
    }
#if DS_QUAD_ROWS
    if (raw || (p.flags & DS_CONV_F_OUT_F32)) halo3_epilogue_rows_f32<true, false>(qp, acc, bz, n0, outHW, shl, coord, coord2, smem + OFF_H + wave_e * EPI_F32_WAVE, s1, s2, 1.0f, lane_e);
#else
    if (raw || (p.flags & DS_CONV_F_OUT_F32)) halo3_epilogue_hp<DS_ACT_NONE, 2, false>(qp, acc, bz, n0, outHW, shl, coord, s1, s2, 1.0f, lane_e);
#endif
    else if (p.act == DS_ACT_GELU) halo3_epilogue<DS_ACT_GELU, true, false>(qp, acc, bz, n0, outHW, shl, coord, s1, s2, 1.0f, lane_e);
#if DS_QUAD_ROWS_BF16
    else halo3_epilogue_rows<DS_ACT_NONE, true, false>(qp, acc, bz, n0, outHW, shl, coord, coord2, smem + OFF_H + wave_e * (16 * 208), s1, s2, 1.0f, lane_e);
#else
    else halo3_epilogue<DS_ACT_NONE, true, false>(qp, acc, bz, n0, outHW, shl, coord, s1, s2, 1.0f, lane_e);
#endif
    __syncthreads();
    if (p.stats_part && !raw) {
        const int parts = gx * gy;
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)bz * parts + by * gx + bx) * 2, lane_e, wave_e);
    }
}

int quad_twl(int W) {
    int twl = 3;
    while ((1 << twl) < W && twl < 5) ++twl;
    return twl;
}

}  // namespace

int ds_conv_quad_halo3_parts(const ds_conv_params* p) {
    const int twl = quad_twl(p->Wo), TW = 1 << twl, TH = BM >> twl;
    return ((p->Ho + TH - 1) / TH) * ((p->Wo + TW - 1) / TW) * (p->cout_pad / BN);
}

int ds_conv_quad_halo3_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv_quad_halo3: bf16 only");
    // (ds_conv_params describes a transposed convolution by its 2x2 phases: KH = KW = 2, stride 1, pad 0, Ho x Wo = the input grid)
    DS_REQUIRE(p->transposed ? (p->KH == 2 && p->KW == 2) : (p->KH == 4 && p->KW == 4 && p->pad_h == 1 && p->pad_w == 1 && p->stride == 2),
               "conv_quad_halo3: Conv2d(4, 2, 1) or ConvTranspose2d(4, 2, 1) only");
    const bool split_in = (p->flags & DS_CONV_F_SPLIT_IN) != 0, out_f32 = (p->flags & DS_CONV_F_OUT_F32) != 0;
    DS_REQUIRE((p->flags & ~(DS_CONV_F_SPLIT_IN | DS_CONV_F_OUT_F32)) == 0 && (!out_f32 || p->act == DS_ACT_NONE), "conv_quad_halo3: unsupported flags %d", p->flags);
    DS_REQUIRE(!split_in || p->C0 % 64 == 0, "conv_quad_halo3: split input needs C0 = 2C with C %% 32 == 0");
    const int plane_chunks = split_in ? (p->C0 / 32) * 3 / 2 : p->C0 / 32;
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0 && ((p->transposed ? 1 : 4) * plane_chunks) % 6 == 0,
               "conv_quad_halo3: single source, Cin %% 32 == 0 and a chunk count that is a multiple of 6 (Cin=%d)", p->C0);
    DS_REQUIRE(p->wk_order == 2 && p->cout_pad % BN == 0, "conv_quad_halo3: quad-packed weights (wk_order = 2), cout_pad %% 96 == 0");
    if (p->transposed) {
        DS_REQUIRE(p->Cout % BN == 0 && p->cout_pad == 4 * p->Cout && p->Ho == p->H && p->Wo == p->W,
                   "conv_quad_halo3: transposed needs Cout %% 96 == 0, cout_pad = 4 Cout and the input grid in Ho / Wo");
    } else {
        DS_REQUIRE(p->H % 2 == 0 && p->W % 2 == 0 && p->Ho == p->H / 2 && p->Wo == p->W / 2, "conv_quad_halo3: strided needs even H, W and Ho = H / 2");
    }
    DS_REQUIRE(!p->gn_ab && !p->gn_part && !p->res && !p->out_nchw_f32 && !p->res_steps, "conv_quad_halo3: no GroupNorm fold, residual or NCHW output");
    const int nchunks = (p->transposed ? 1 : 4) * plane_chunks;
    DS_REQUIRE(p->ksplit <= 1 || (p->slab && nchunks % p->ksplit == 0 && (nchunks / p->ksplit) % 6 == 0),
               "conv_quad_halo3: ksplit=%d needs a slab and slices of a multiple of 6 chunks (%d chunks)", p->ksplit, nchunks);
    const long long oHW = (long long)p->Ho * p->Wo * (p->transposed ? 4 : 1);
    DS_REQUIRE((long long)p->H * p->W * p->C0 * 2 < (1ll << 31) && oHW * p->out_C * (out_f32 ? 4 : 2) < (1ll << 31) &&
                   (long long)(p->transposed ? 1 : 4) * plane_chunks * 4 * p->cout_pad * 64 < (1ll << 31),
               "conv_quad_halo3: one sample / the packed weights must stay below 2 GiB (32-bit buffer offsets)");
    const int twl = quad_twl(p->Wo), TW = 1 << twl, TH = BM >> twl;
    dim3 grid(((p->Ho + TH - 1) / TH) * ((p->Wo + TW - 1) / TW), p->cout_pad / BN, p->B * (p->ksplit > 1 ? p->ksplit : 1));
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_HALO);
        ds_conv_bounds_table(*p, DS_K_CONV_HALO, grid.x * grid.y, &h.t);
        h.set(DS_BX_W, p->wpk, (long long)(p->transposed ? 1 : 4) * plane_chunks * 4 * p->cout_pad * 64);
        if (out_f32) h.set(DS_BX_OUT, p->out, (long long)p->B * oHW * p->out_C * 4);
        h.publish(st);
    }
#endif
    if (twl == 5) {
        DS_SET_MAX_LDS(conv_quad_halo3_kernel<5>, QLDS_BYTES, "conv_quad_halo3<32>");
        hipLaunchKernelGGL(conv_quad_halo3_kernel<5>, grid, dim3(NT), QLDS_BYTES, st, *p);
    } else if (twl == 4) {
        DS_SET_MAX_LDS(conv_quad_halo3_kernel<4>, QLDS_BYTES, "conv_quad_halo3<16>");
        hipLaunchKernelGGL(conv_quad_halo3_kernel<4>, grid, dim3(NT), QLDS_BYTES, st, *p);
    } else {
        DS_SET_MAX_LDS(conv_quad_halo3_kernel<3>, QLDS_BYTES, "conv_quad_halo3<8>");
        hipLaunchKernelGGL(conv_quad_halo3_kernel<3>, grid, dim3(NT), QLDS_BYTES, st, *p);
    }
    DS_CHECK_LAUNCH("conv_quad_halo3");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_quad(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
