// 3x3 stride-1 pad-1 convolution, bf16, LDS input halo — the conv3x3_halo2 pipeline on 16x16x32 MFMAs (gfx950).
//
// Same block tile (256 px x 96 ch, 4 waves of 64 px x 96 ch, two blocks per CU), same data flow and the same hand-placed
// step as conv3x3_halo2.hip (chunk-major weights through a 3-slot register ring into a 3-buffer LDS ring, input halo of a
// 32-channel chunk double-buffered in LDS, buffer loads with scalar offsets, counted lgkmcnt + raw s_barrier, fused 1x1
// res_conv steps in front of the nine-tap chunks).  What differs:
//
//   * v_mfma_f32_16x16x32_bf16 instead of 32x32x16.  halo2's K loop is pipe-bound (92-95 % MFMA-busy) at a clock the chip
//     lowers to 1.5 GHz under it; the 16x16x32 shape sustains a higher clock for the same FLOPs (MI355X_MICROARCH.md:
//     1.12-1.14x with operands re-read from LDS).  One MFMA consumes the whole 32-channel chunk of a tap, so a step is
//     24 MFMAs of 16 cycles on 4 pixel fragments x 6 weight fragments — ten ds_read_b128 per step, as before.
//   * LDS images with 64-byte rows (no pad) and an XOR swizzle instead of the 80-byte pitch: a 16-row x 64-byte fragment
//     read puts lanes {0-3, 12-15} of one 16-byte column and lanes {4-11} of the next into one bank group, which any odd
//     pitch makes a 2-way conflict.  Halo pixel hp keeps its 16-byte quarters at (q ^ 2*bit2(hp)); halo rows are padded
//     to a multiple of 4 pixels so that a tap shift changes bit2 by a per-lane constant: a fragment address is
//     (base_i ^ mask(tap)) + immediate, one v_xor per fragment read.  Weight rows are swizzled by W_SWZ(row) = (-((row >> 3) & 3)) & 3 and
//     MFMA row m of weight fragment j is channel 24*(m>>2) + 4*j + (m&3): lane group g then accumulates channels
//     24g .. 24g+23 of its pixel — 48 contiguous output bytes per lane, no cross-lane permute in the epilogue (halo2 needs
//     one v_permlane32_swap + s_nop per accumulator register).  All reads and writes are conflict-free (checked by
//     enumeration for the three tile shapes, every tap and every lane group).
//   * the kernel is templated on the tile width (32 / 16 / 8 px): halo pitch, tap offsets and trip counts are immediates.
#include "common.hpp"
#ifndef DS_STAMP
#define DS_STAMP 0   // diagnostic build: per-wave s_memtime / s_memrealtime stamps around prologue, K loop and epilogue -> p.slab (8 longs per wave)
#endif
#ifndef DS_NGROUP
#define DS_NGROUP 2   // N-blocks per group of the block order (0: N-block fastest over the whole layer, the order up to r03)
#endif
#if DS_BOUNDS
void ds_conv_bounds_table(const ds_conv_params& p, int kernel, int stats_parts, ds_bx* out);   // conv_igemm.hip
#endif

#include "conv_halo3_common.hpp"

#ifndef DS_EPI_ROWS_NORES
#define DS_EPI_ROWS_NORES 0      // (with the 24-channel lane map the staged form won 1..5 % here; with 64-byte runs from the registers it does not)
#endif
#ifndef DS_EPI_ROWS
#define DS_EPI_ROWS 1      // line-sized stores through an LDS tile (halo3_epilogue_rows); 0: the register-only epilogue (A/B)
#endif

namespace {

// GELU table of the bf16 epilogue (conv_halo3_common.hpp: gelu_tab8): T(a) = a Phi(-a) at the midpoints of the bf16 buckets of [2^-12, 8),
// filled once per device by the launcher (erfc in double precision on the host)
__device__ __attribute__((aligned(16))) float g_gelu_lut[GELU_TAB_N];

// HP = the split-precision instantiation (ds_conv_params.flags != 0): split input planes and / or split or fp32 output, no fused
// res_conv phase; a separate instantiation so that the bf16 kernel's register allocation (249-253 of 256, no spills) is untouched.
#ifdef DS_FORCE_VGPRS      // diagnostic: a register budget below what the kernel needs forces spills into scratch
#define DS_VGPR_ATTR __attribute__((amdgpu_num_vgpr(DS_FORCE_VGPRS)))
#else
#define DS_VGPR_ATTR
#endif
// PAIR (r05, split-precision launches of the 8-wide tile only): TWO samples per block.  The 8 x 32 tile of the deepest level is one sample
// at 256 x 64 latents (32 x 8 images) but HALF empty at the reference's own 128 x 64 (16 x 8 images: text2sound.py:84) — conv3x3_halo3<3>
// took the same 414 us per launch at both sizes, 12 % of a 128 x 64 step.  Rows 0 .. 15 of the tile are sample 2z, rows 16 .. 31 sample
// 2z + 1 (waves 0, 1 / waves 2, 3); each sample keeps its own zero rows above and below in the halo (18 + 18 halo rows), its own GroupNorm
// factor and shift table, its own statistics partial.  Whole-K launches only.
// ONE block per CU for PAIR: its parent sits at exactly 256 registers, and the handful of values a second sample adds spilled the halo
// offsets inside the K loop (101 - 152 spilled registers in every arrangement tried: the block then took more than twice its time and the
// launch was 2 % SLOWER than the half-empty tiles).  With 512 registers nothing spills; a lone block runs its loop at 0.7 x the paired
// rate, and half of these launches have only 256 blocks anyway (4 N-blocks x 64 sample pairs at batch 128).
template <int TWL, bool HP, bool PAIR = false>
__global__ __launch_bounds__(NT, PAIR ? 1 : DS_MINBLK) DS_VGPR_ATTR void conv3x3_halo3_kernel(const ds_conv_params p, const int lut_on) {
    static_assert(!PAIR || (HP && TWL == 3), "two samples per block: the 8 x 32 tile of the split-precision instantiation");
    using G = HG<TWL>;
    constexpr int TW = G::TW, TH = G::TH, HCP = G::HCP, NPX = G::NPX, H_IT = G::H_IT, HH0 = G::HH0, HH1 = G::HH1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const shl = reinterpret_cast<float*>(smem + OFF_SHL);
    float* const red = reinterpret_cast<float*>(smem + OFF_B);      // reused after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long st_k0 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    const int m = lane & 15, q = lane >> 4;
    const int tiles_w = (p.W + TW - 1) >> TWL;
    // XCD-chunked block order.  Hardware block L (x fastest) runs on XCD L % 8, each with its own L2.  Logical work item
    // w = (L % 8) * (n / 8) + L / 8, decoded with the N-block fastest, then the tile, then the sample: the N-blocks of one tile
    // (same input halo) and the neighbouring tiles of one sample (shared halo rows) are consecutive on ONE XCD, so every input
    // byte comes from beyond L2 once.  In plain order the N-blocks of the 32 x 8 level (one tile per sample) sat on 8 different
    // XCDs and the tiles of a row on different ones.  (Streamed data served from L2 is also the largest term of the energy per
    // MFMA, which sets the clock this loop runs at.)
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    int wid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
#ifndef DS_HALO3_NOXCD
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
#endif
#if DS_NGROUP
    // r04: N-blocks in groups of DS_NGROUP (2) fastest, then the tile and the sample, then the N-group — the blocks resident on an XCD at one
    // time then stream the SAME two N-blocks' weights (shared through its L2) instead of all of the layer's: fabric reads of the 64 x 16 /
    // 32 x 8 launches 2.44 -> 1.84 / 0.61 -> 0.45 GB (PMC FETCH_SIZE), step time unchanged (717.7 vs 717.4 steps/s, same box; groups of 1
    // lose the input's L2 reuse: -0.7 %)
    int by, bxz;
    if (gy % DS_NGROUP == 0 && gy > DS_NGROUP) {
        const int lo = wid % DS_NGROUP, t = wid / DS_NGROUP, nxz = gx * (int)gridDim.z;
        bxz = t % nxz;
        by = (t / nxz) * DS_NGROUP + lo;
    } else {
        by = wid % gy;
        bxz = wid / gy;
    }
    const int bx = bxz % gx, bz = bxz / gx;
#else
    const int by = wid % gy, bxz = wid / gy, bx = bxz % gx, bz = bxz / gx;
#endif
    const int th = bx / tiles_w, tw = bx - th * tiles_w;
    const int h0 = th * TH, w0 = tw * TW;
    // split-K (small batches at the small-spatial levels: too few blocks for 256 CUs): blockIdx.z = sample * ksplit + K slice; a slice
    // accumulates its share of the channel chunks and stores raw fp32 partial sums to p.slab, ds_conv_splitk_reduce adds the slices and
    // runs the epilogue (bias / fold / activation / residual / statistics)
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int b = (PAIR ? 2 : 1) * (bz / ksplit), kz = bz - (bz / ksplit) * ksplit, n0 = by * BN;      // (PAIR: the block's FIRST sample)
    const bool has2 = PAIR && b + 1 < p.B;
    // split-precision input (flags & DS_CONV_F_SPLIT_IN): src0 holds 2C bf16 channels = the hi plane then the lo plane of a C-channel
    // fp32 tensor; the packed weights hold, per 32-channel source chunk c, the three virtual chunks [W_hi_c | W_lo_c | W_hi_c]
    // (engine.split3_weight).  The K loop multiplies the hi plane's chunk c with the first two — ONE staged halo serves both (r04: the
    // [W_hi | W_hi | W_lo] order staged every hi chunk twice, 1.5 x the plane bytes per block) — and the lo plane's chunk c with the
    // third: x*w ~ x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on bf16 MFMAs
    constexpr bool split_in = HP;      // (the launcher requires DS_CONV_F_SPLIT_IN for every split-precision launch)
    const int Cin = p.C0, NSRC = Cin >> 5;                          // chunks the source holds
    const int NCC = (split_in ? NSRC + (NSRC >> 1) : NSRC) / ksplit;      // chunks of this block's K loop
    const int cc_lo = kz * NCC;                                           // first chunk of the K slice (0 without split-K)
    const int nsteps = NCC * 9;
    // source chunk of virtual chunk cc_lo + cc: plain input — itself; split input — virtual chunk 3c + j reads the hi plane's chunk c
    // (j = 0, 1) or the lo plane's (j = 2: NSRC / 2 + c); a K slice starts at a multiple of three
    auto src_chunk = [&](int cc) {
        const int v = cc_lo + cc;
        if (!split_in) return v;
        const int c = v / 3, j = v - 3 * c;
        return j == 2 ? (NSRC >> 1) + c : c;
    };

    // ---- resource descriptors (wave-uniform) and per-thread offsets, all fixed for the whole kernel
    const int NR = (HP || ksplit > 1) ? 0 : p.res_steps, R0 = p.res_C0 >> 5;
    const char* const wbase = reinterpret_cast<const char*>(p.wpk);
    const unsigned wbytes = (unsigned)(NCC * ksplit * 9 + p.res_steps) * p.cout_pad * 64;      // (NCC counts the virtual chunks of a split input)
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), (short)0, (int)wbytes, 0x00020000);

    // Linear chunk order of a block: first the NR one-step chunks of the fused res_conv (32 channels of res_src0, then of
    // res_src1 placed at its pad offset: pad_and_concat, components:210-249), then the NCC nine-tap chunks of the 3x3 input.
    // hvo = byte offset of this thread's 16 B of chunk 0 inside the current source's sample (VOFF_NONE: outside the image,
    // a pad column, or not needed — the range check returns zeros without touching memory).
    unsigned hvo[H_IT];
    rsrc_t rs_h;
    const char* hbase;
    int h_buf = DS_BX_SRC0;
    auto use_source = [&](const void* ptr, int sH, int sW, int sC, int offh, int offw, bool centre_only, int bounds_buf) {
        hbase = reinterpret_cast<const char*>(ptr) + (size_t)b * sH * sW * sC * 2;
        rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(hbase), (short)0, (int)((unsigned)sH * sW * sC * 2) * (has2 ? 2 : 1), 0x00020000);
        h_buf = bounds_buf;
#pragma unroll
        for (int it = 0; it < H_IT; ++it) {
            const int slot = tid + it * NT, hp = slot >> 2, dq = slot & 3;
            int hr = hp / HCP;                                           // HCP is a compile-time constant: multiply + shift
            const int hc = hp - hr * HCP;
            // PAIR: halo rows 0 .. 17 = image rows -1 .. 16 of the first sample, 18 .. 35 the same of the second (H <= 16)
            const int smp = PAIR ? (int)(hr >= 18) : 0;
            if constexpr (PAIR) hr -= 18 * smp;
            // (arithmetic, not nested ifs: those become exec-masked regions per slot; an offset with bit 31 set = VOFF_NONE)
            const int hi = h0 + hr - 1 - offh, wi = w0 + hc - 1 - offw;
            const unsigned ring = (unsigned)(hr < 1) | (unsigned)(hr > TH) | (unsigned)(hc < 1) | (unsigned)(hc > TW);   // a 1x1 never reads the halo ring
            const unsigned bad = (unsigned)(hp >= (PAIR ? 36 * HCP : NPX)) | (unsigned)(hc >= TW + 2) | ((unsigned)centre_only & ring) |
                                 (unsigned)((unsigned)hi >= (unsigned)sH) | (unsigned)((unsigned)wi >= (unsigned)sW) | (unsigned)(smp && !has2);
            hvo[it] = (((unsigned)(((smp * sH + hi) * sW + wi) * sC + dq * 8) * 2u) & 0x7fffffffu) | (bad << 31);
        }
    };
    unsigned h_so = 0;           // scalar byte offset of the chunk inside a pixel's channels
    auto set_res_src = [&](int r) {   // select res chunk r (0 <= r < NR) or, for r == NR, the first 3x3 chunk; consecutive r only
        if (r == R0 && r < NR) use_source(p.res_src1, p.res_H1, p.res_W1, p.res_C1, p.res_off_h1, p.res_off_w1, true, DS_BX_AUX1);
        if (r == NR) use_source(p.src0, p.H, p.W, Cin, 0, 0, false, DS_BX_SRC0);
        h_so = r == NR ? (unsigned)src_chunk(0) * 64u : (unsigned)(r < R0 ? r : r - R0) * 64u;
    };
    if (NR > 0) use_source(p.res_src0, p.H, p.W, p.res_C0, 0, 0, true, DS_BX_AUX0);
    else use_source(p.src0, p.H, p.W, Cin, 0, 0, false, DS_BX_SRC0);

    // LDS store offsets.  Halo: slot -> pixel hp = slot >> 2, quarter dq = slot & 3 at hp * 64 + ((dq ^ 2 * bit2(hp)) * 16); the
    // 64 pixels of an iteration leave bit2 alone, so iteration `it` is the same offset + it * 4096.
    const int lds_h = OFF_H + (tid >> 2) * PSTR + (((tid & 3) ^ (((tid >> 4) & 1) << 1)) << 4);
    // Weights: piece t -> row t >> 2 (+ 64 for the second round: threads 0..127), quarter swizzled by W_SWZ(row)
    const int wr0 = tid >> 2, wr1 = 64 + (tid >> 2);
    const int bst0 = OFF_B + wr0 * PSTR + (((tid & 3) ^ W_SWZ(wr0)) << 4);
    const int bst1 = tid < BN * 4 - NT ? OFF_B + wr1 * PSTR + (((tid & 3) ^ W_SWZ(wr1)) << 4) : OFF_B + B_BYTES;
    const unsigned wvo0 = (unsigned)tid * 16u, wvo1 = tid < BN * 4 - NT ? (unsigned)(tid + NT) * 16u : VOFF_NONE;
    const unsigned wstep = (unsigned)p.cout_pad * 64u;                                  // bytes per K step
    // wpk = [res tiles (p.res_steps)][3x3 tiles]: a fused launch starts at the res tiles, a K slice at its first chunk
    const unsigned w_first = ((unsigned)(p.res_steps - NR + cc_lo * 9) * p.cout_pad + n0) * 64u, w_last = w_first + (unsigned)(nsteps + NR - 1) * wstep;
    unsigned w_pf = w_first;     // scalar offset of the next weight tile to fetch (clamped at the last real step: tail loads are dummies)

    u32x4 rb[3][2], rh[H_IT];    // rh: the 3x3 chunks refill the halo in two halves through rh[0 .. HH0); the 1-step res chunks use all of it
    auto load_b = [&](auto slotc) {
        constexpr int sl = decltype(slotc)::value;
        rb[sl][0] = buf_ld16(rs_w, wbase, wvo0, w_pf, DS_BX_W);
        rb[sl][1] = buf_ld16(rs_w, wbase, wvo1, w_pf, DS_BX_W);
        const unsigned nx = w_pf + wstep;
        w_pf = nx < w_last ? nx : w_last;
    };
    auto store_b = [&](auto slotc, auto bufc) {
        constexpr int sl = decltype(slotc)::value, buf = decltype(bufc)::value;
        *reinterpret_cast<u32x4*>(smem + buf * B_STRIDE + bst0) = rb[sl][0];
        *reinterpret_cast<u32x4*>(smem + buf * B_STRIDE + bst1) = rb[sl][1];
    };
    auto load_halo_to = [&](u32x4* dst, auto halfc) {   // the source / chunk is whatever was selected last: rs_h / hvo / h_so
        constexpr int half = decltype(halfc)::value, n = half ? HH1 : HH0;
#pragma unroll
        for (int k = 0; k < n; ++k) dst[k] = buf_ld16(rs_h, hbase, hvo[half * HH0 + k], h_so, h_buf);
    };
    auto load_halo = [&](auto halfc) { load_halo_to(rh, halfc); };
    auto store_halo_from = [&](const u32x4* src, auto bufc, auto halfc) {
        constexpr int buf = decltype(bufc)::value, half = decltype(halfc)::value, n = half ? HH1 : HH0;
#pragma unroll
        for (int k = 0; k < n; ++k) *reinterpret_cast<u32x4*>(smem + lds_h + buf * G::HB + (half * HH0 + k) * 64 * PSTR) = src[k];
    };
    auto store_halo = [&](auto bufc, auto halfc) { store_halo_from(rh, bufc, halfc); };

    // ---- per-lane fragment bases.  Pixel of (tile i, lane m): tile-local (row, col)
    auto tile_rc = [&](int i, int& row_l, int& col_l) {
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + m; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = m; }
        else { row_l = 8 * wave + i + 4 * (m >> 3); col_l = m & 7; }      // rows (i, i + 4): conflict-free with the 12-pixel pitch
    };
    int xb[XT];                  // address of tap (0, 0) in halo buffer 0
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        int row_l, col_l;
        tile_rc(i, row_l, col_l);
        const int hp0 = (row_l + (PAIR ? 2 * (row_l >> 4) : 0)) * HCP + col_l;      // (PAIR: the second sample's rows sit two halo rows lower)
        xb[i] = OFF_H + hp0 * PSTR + ((q ^ (((hp0 >> 2) & 1) << 1)) << 4);
    }
    // bit2(hp0 + ty * HCP + tx) = bit2(hp0) ^ (ty == 1) ^ carry(tx), carry(1) = (m & 3) == 3, carry(2) = (m & 3) >= 2 (the column of
    // every tile is m modulo 4; HCP / 4 is odd for the three shapes): XOR masks on address bit 5
    static_assert(((HCP >> 2) & 1) == 1 && ((2 * HCP >> 2) & 1) == 0, "tap-row swizzle flips assume HCP / 4 odd");
    const int xm1 = ((m & 3) == 3) << 5, xm2 = ((m & 3) >= 2) << 5;
    const int xm1n = xm1 ^ 32, xm2n = xm2 ^ 32;
    const int bw = OFF_B + W_ROW0(m) * PSTR + ((q ^ ((-(m >> 2)) & 3)) << 4);      // + EPI_CH(j) rows for tile j

    bf16x8 fx[2][XT], fw[WT];
    auto read_x = [&](auto setc, auto tyc, auto txc, int imm) {
        constexpr int set = decltype(setc)::value, ty = decltype(tyc)::value, tx = decltype(txc)::value;
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            int a = xb[i];
            if constexpr (HP) {
                // (the split-precision instantiation is three registers over its budget with the hi-chunk reuse: the two negated masks are an
                // extra XOR at six of 36 fragment reads per chunk instead of two live registers — opaque, or loop-invariant code motion
                // brings the registers back)
                if constexpr (tx == 1) a ^= xm1;
                if constexpr (tx == 2) a ^= xm2;
                if constexpr (ty == 1) {
                    if constexpr (tx != 0) asm volatile("" : "+v"(a));
                    a ^= 32;
                }
            } else {
                if constexpr (tx == 0 && ty == 1) a ^= 32;
                if constexpr (tx == 1) a ^= (ty == 1 ? xm1n : xm1);
                if constexpr (tx == 2) a ^= (ty == 1 ? xm2n : xm2);
            }
            fx[set][i] = *reinterpret_cast<const bf16x8*>(smem + a + (ty * HCP + tx) * PSTR + imm);
        }
    };
#ifndef DS_HALO3_ABL
#define DS_HALO3_ABL 0   // timing experiments only (wrong results): bit0 weight fragments read once, bit1 pixel fragments read once, bit2 no per-step barrier
#endif
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

    // ---- prologue: ONE memory round trip.  The producer's statistics partials are requested before anything else (their float64
    // reduction then runs while the halo and the weight tiles are still in flight: loads retire in order), then the fold-table entries
    // this thread combines, then the big loads.
    GnPartialLoads gnl;
    const bool raw = ksplit > 1;                     // K slice: zero shift table, factor 1, no statistics
    if (p.gn_part && !raw) gn_partials_issue(p.gn_part, p.gn_parts, b, gnl);
    const bool fold = !raw && (p.gn_ab != nullptr || p.gn_part != nullptr);
    const int ncls = fold ? p.ncls : 1;
    constexpr int ST_IT = (10 * BN + NT - 1) / NT;     // 4 shift-table entries per thread at most (row 9 stays zero)
    float t1v[ST_IT], t2v[ST_IT], rbv[ST_IT];
#if DS_BOUNDS
#pragma unroll
    for (int k = 0; k < ST_IT; ++k) {
        const int e = tid + k * NT, cls = e / BN, n = n0 + e - cls * BN;
        t1v[k] = 0.f;
        t2v[k] = 0.f;
        if (e < ncls * BN && n < p.Cout) {
            if (fold) {
                t1v[k] = DS_LD(float, p.fold_t1 + cls * p.Cout + n, DS_BX_T1);
                t2v[k] = DS_LD(float, p.fold_t2 + cls * p.Cout + n, DS_BX_T2);
            } else if (p.bias && !raw) t1v[k] = DS_LD(float, p.bias + n, DS_BX_BIAS);
            if (NR > 0 && p.res_bias) t1v[k] += DS_LD(float, p.res_bias + n, DS_BX_AUX2);
        }
        rbv[k] = 0.f;
    }
#else
    // Range-checked buffer loads with arithmetic out-of-range offsets (bit 31): written as `if (valid) t = load` every entry became an
    // exec-masked region with its own s_waitcnt vmcnt(0) — eight serial L2 round trips in front of the halo request.
    {
        auto tab = [&](const float* ptr, int n) {      // (an absent table: zero records over any valid address — every load returns 0)
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ptr ? reinterpret_cast<const char*>(ptr) : wbase), (short)0, ptr ? n * 4 : 0, 0x00020000);
        };
        const bool use_bias = !fold && p.bias && !raw;
        const rsrc_t rs_t1 = tab(fold ? p.fold_t1 : (use_bias ? p.bias : nullptr), fold ? ncls * p.Cout : p.Cout);
        const rsrc_t rs_t2 = tab(fold ? p.fold_t2 : nullptr, ncls * p.Cout);
        const rsrc_t rs_rb = tab((NR > 0 && p.res_bias) ? p.res_bias : nullptr, p.Cout);
#pragma unroll
        for (int k = 0; k < ST_IT; ++k) {
            const int e = tid + k * NT, cls = e / BN, n = n0 + e - cls * BN;
            const unsigned bad = ((unsigned)(e >= ncls * BN) | (unsigned)(n >= p.Cout)) << 31;
            const unsigned o = (unsigned)((fold ? cls * p.Cout : 0) + n) * 4u;
            t1v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t1, (int)((o & 0x7fffffffu) | bad), 0, 0));
            t2v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t2, (int)((o & 0x7fffffffu) | bad), 0, 0));
            // (the fused res_conv's bias: its own register until the table is written — an add here would wait for both loads on the spot)
            rbv[k] = NR > 0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rb, (int)((((unsigned)n * 4u) & 0x7fffffffu) | bad), 0, 0)) : 0.f;
        }
    }
#endif
    const bool use_lut = !HP && G::LUT && lut_on && p.act == DS_ACT_GELU && !raw;
    static_assert(GELU_TAB_N * 4 <= 2 * NT * 16 && GELU_TAB_N % 4 == 0, "the table is staged as two 16-byte vectors per thread");
    u32x4 lutv = {0u, 0u, 0u, 0u}, lutv2 = {0u, 0u, 0u, 0u};
    if (use_lut) {
        lutv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(g_gelu_lut) + tid * 16);                // 1920 x 4 B = 480 x 16 B
        // (unconditional, clamped: a load inside `if (tid < 224)` is an exec-masked region that ends in s_waitcnt vmcnt(0))
        lutv2 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(g_gelu_lut) + (NT + min(tid, GELU_TAB_N / 4 - NT - 1)) * 16);
    }
    const long st_p1 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;    // setup + small loads issued
    u32x4 rh2[HH0];
    h_so = NR > 0 ? 0u : (unsigned)src_chunk(0) * 64u;
    load_halo_to(rh2, I0{});
    load_halo(I1{});
    load_b(I0{});
    load_b(I1{});
    float gn_a = 1.f, gn_am = 0.f;
    if (raw) {
    } else if (p.gn_part) gn_partials_finish(gnl, p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, gn_a, gn_am);
    else if (p.gn_ab) {
        gn_a = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
        gn_am = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
    }
    if constexpr (PAIR) {
        // the second sample's GroupNorm factor and mean: reduced HERE and parked in the 64 bytes the launcher adds behind the kernel's LDS
        // image (kept from the two-blocks-per-CU attempts, where anything carried through the K loop spilled: it costs nothing)
        const int b1 = has2 ? b + 1 : b;
        float a1 = 1.f, am1 = 0.f;
        if (p.gn_part) gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b1, a1, am1);
        else if (p.gn_ab) {
            a1 = DS_LD(float, p.gn_ab + 2 * b1, DS_BX_GNAB);
            am1 = DS_LD(float, p.gn_ab + 2 * b1 + 1, DS_BX_GNAB);
        }
        if (tid == 0) {
            reinterpret_cast<float*>(smem + G::LDS)[0] = a1;
            reinterpret_cast<float*>(smem + G::LDS)[1] = am1;
        }
    }
#pragma unroll
    for (int k = 0; k < ST_IT; ++k) {
        const int e = tid + k * NT;
        if (e < 10 * BN) shl[e] = (t1v[k] + rbv[k]) - gn_am * t2v[k];      // entries beyond the ncls real rows are zeros
    }
    const long st_p2 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;    // statistics reduced, shift table written
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (use_lut) {
        *reinterpret_cast<u32x4*>(smem + G::OFF_LUT + tid * 16) = lutv;
        if (tid < GELU_TAB_N / 4 - NT) *reinterpret_cast<u32x4*>(smem + G::OFF_LUT + (NT + tid) * 16) = lutv2;
    }
    store_halo_from(rh2, I0{}, I0{});
    store_halo(I0{}, I1{});
    store_b(I0{}, I0{});
    store_b(I1{}, I1{});
    load_b(I2{});
    load_b(I0{});
    load_b(I1{});
    __syncthreads();

    // ---- fused res_conv: NR one-step chunks at the centre tap (NR % 3 == 0: the weight ring is back at phase 0 when the
    // nine-tap chunks start).  Each step pays one LDS read latency (the halo buffer of the next step is being written during
    // the step): NR is 3 .. 24 against 27 .. 216 nine-tap steps.  The last res step stages the first 3x3 chunk's halo.
    if constexpr (!HP) if (NR > 0) {
        set_res_src(1);
        load_halo_to(rh, I0{});
        load_halo_to(rh + HH0, I1{});
        auto res_step = [&](auto hbc, auto phc, int r) {
            constexpr int hb = decltype(hbc)::value, ph = decltype(phc)::value, rs = (ph + 2) % 3;
            read_x(I0{}, I1{}, I1{}, hb * G::HB);
#pragma unroll
            for (int j = 0; j < WT; ++j) read_w(j, ph * B_STRIDE);
            __builtin_amdgcn_sched_barrier(0);
            store_halo_from(rh, std::integral_constant<int, hb ^ 1>{}, I0{});          // chunk r + 1 (requested one step ago)
            store_halo_from(rh + HH0, std::integral_constant<int, hb ^ 1>{}, I1{});
            store_b(std::integral_constant<int, rs>{}, std::integral_constant<int, rs>{});
            __builtin_amdgcn_sched_barrier(0);
            load_b(std::integral_constant<int, rs>{});
            if (r + 2 <= NR) {                         // r + 2 == NR: the first 3x3 chunk (full halo of the 3x3 input)
                set_res_src(r + 2);
                __builtin_amdgcn_sched_barrier(0);
                load_halo_to(rh, I0{});
                load_halo_to(rh + HH0, I1{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < WT; ++j) mma_j(I0{}, j);
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
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= inv_a;
    }
    // bf16: halo buffer of the first 3x3 chunk = fragment set of its first step (an odd number of res steps in front);
    // split precision: fragment set 1 for the peeled triple of an odd number of source chunks (its halo is in buffer 0)
    const int par = HP ? 0 : (NR & 1);
    if (par) read_x(I1{}, I0{}, I0{}, G::HB);
    else read_x(I0{}, I0{}, I0{}, 0);
#pragma unroll
    for (int j = 0; j < WT; ++j) read_w(j, 0);

    // ---- main loop: chunks x 9 taps.  hbuf (halo double buffer = fragment set parity) and every ring index are compile-time.
    // keepc = 1: the NEXT chunk multiplies the same staged halo (a hi chunk's second weight set): nothing is prefetched, the halo buffers do
    // not swap.  fsc = parity of the fragment set this chunk's first step reads (chunks of nine steps flip it; with halo buffers that do not
    // swap every chunk it is no longer the buffer's parity).
    auto chunk = [&](auto hbufc, auto keepc, auto fsc, int cc) {
        constexpr int hbuf = decltype(hbufc)::value, KEEP = decltype(keepc)::value, FS = decltype(fsc)::value;
        if constexpr (!KEEP) h_so = (unsigned)src_chunk(cc + 1 < NCC ? cc + 1 : cc) * 64u;      // chunk prefetched during this one (a dummy re-read at the end)
        auto step = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int rs = (tap + 2) % 3;                  // ring slot stored this step (tile s + 2), then refilled with tile s + 5
            constexpr int cur = (tap + FS) & 1;
            constexpr int ntap = (tap + 1) % 9, nty = ntap / 3, ntx = ntap % 3, nhb = (tap == 8 && !KEEP) ? (hbuf ^ 1) : hbuf;
            constexpr int nW = 2 + (KEEP ? 0 : (tap == 3 ? HH0 : (tap == 7 ? HH1 : 0))), nV = 2 + (KEEP ? 0 : (tap == 1 ? HH0 : (tap == 4 ? HH1 : 0)));
            if constexpr (tap == 3 && !KEEP) store_halo(std::integral_constant<int, hbuf ^ 1>{}, I0{});
            if constexpr (tap == 7 && !KEEP) store_halo(std::integral_constant<int, hbuf ^ 1>{}, I1{});
            store_b(std::integral_constant<int, rs>{}, std::integral_constant<int, rs>{});
            load_b(std::integral_constant<int, rs>{});
            if constexpr (tap == 1 && !KEEP) load_halo(I0{});
            if constexpr (tap == 4 && !KEEP) load_halo(I1{});
            // this step's 24 MFMAs, weight fragment j re-read (for the next step) as soon as its four MFMAs are issued
            mma_j(std::integral_constant<int, (DS_HALO3_ABL & 2) ? 0 : cur>{}, 0);
            if constexpr (!(DS_HALO3_ABL & 1)) read_w(0, ((tap + 1) % 3) * B_STRIDE);
            if constexpr (!(DS_HALO3_ABL & 2))
                read_x(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, nty>{}, std::integral_constant<int, ntx>{}, nhb * G::HB);
#pragma unroll
            for (int j = 1; j < WT; ++j) {
                mma_j(std::integral_constant<int, (DS_HALO3_ABL & 2) ? 0 : cur>{}, j);
                if constexpr (!(DS_HALO3_ABL & 1)) read_w(j, ((tap + 1) % 3) * B_STRIDE);
            }
            // one LDS / VMEM instruction per MFMA gap: LDS writes first (the step's barrier waits for them and for nothing else),
            // then the loads for later steps, then the ten fragment reads of the next step
#pragma unroll
            for (int k = 0; k < nW; ++k) { SGB(SG_MFMA, 1); SGB(SG_DSW, 1); }
#pragma unroll
            for (int k = 0; k < nV; ++k) { SGB(SG_MFMA, 1); SGB(SG_VMEM, 1); }
            // MFMA index reached so far: L0 = nW + nV (4, 7 or 8).  Reads: w0 (needs MFMAs 0..3), x0..x3, then w_j after MFMA 4j + 3.
            constexpr int L0 = nW + nV;
#pragma unroll
            for (int k = 0; k < 5; ++k) { SGB(SG_MFMA, 1); SGB(SG_DSR, 1); }
            constexpr int L1 = L0 + 5;                         // 9, 12 or 13 MFMAs placed
            // w1..w5: w_j goes after MFMA number max(4j + 4, L1 + j - 1) (1-based count of MFMAs placed before it)
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
            // all LDS writes of this step precede its ten fragment reads (program order = completion order): waiting until at
            // most ten LDS operations are outstanding retires the writes and leaves the reads in flight across the barrier
#if DS_BOUNDS || defined(DS_LGKM0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the checker's extra code may reorder the step: no counted wait in this build
#else
            asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
#endif
            if constexpr (!(DS_HALO3_ABL & 4)) __builtin_amdgcn_s_barrier();      // (ABL bit2: no per-step barrier — wrong results, an upper bound)
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
    // (an if / else between the two instantiations inside the loop makes the register allocator keep two copies of the
    // accumulators at the join and spill; an odd start is peeled instead)
    long st_t0 = 0, st_r0 = 0;
    if constexpr (DS_STAMP) {
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) alone: the stamps must not sit in front of the loop's counted LDS waits
    }
    if constexpr (HP) {
        // per source chunk: (hi, W_hi) keeping the halo, (hi, W_lo) while the lo chunk's halo arrives, (lo, W_hi) while the next hi chunk's
        // does.  Two halo swaps and three fragment-set flips per source chunk: the fragment-set parity alternates from one source chunk to the
        // next, so the loop walks PAIRS of source chunks; an odd count peels one triple in FRONT of the loop (started from fragment set 1 —
        // a branch inside the loop makes the register allocator keep two copies of the accumulators at the join and spill).
        int cc = 0;
        for (; cc + 6 <= NCC; cc += 6) {
            chunk(I0{}, I1{}, I0{}, cc);
            chunk(I0{}, I0{}, I1{}, cc + 1);
            chunk(I1{}, I0{}, I0{}, cc + 2);
            chunk(I0{}, I1{}, I1{}, cc + 3);
            chunk(I0{}, I0{}, I0{}, cc + 4);
            chunk(I1{}, I0{}, I1{}, cc + 5);
        }
        if (cc < NCC) {                      // an odd number of source chunks: one more triple behind the loop
            chunk(I0{}, I1{}, I0{}, cc);
            chunk(I0{}, I0{}, I1{}, cc + 1);
            chunk(I1{}, I0{}, I0{}, cc + 2);
        }
    } else {
        int cc0 = 0;
        if (par) {
            chunk(I1{}, I0{}, I1{}, 0);
            cc0 = 1;
        }
        for (int cc = cc0; cc < NCC; cc += 2) {
            chunk(I0{}, I0{}, I0{}, cc);
            if (cc + 1 < NCC) chunk(I1{}, I0{}, I1{}, cc + 1);
        }
    }

    // The last step's ten fragment reads have no consumer; they must stay: the counted wait in front of its barrier retires the step's LDS
    // writes only if ten reads follow them (a triple behind the loop made them dead code, and tests/test_isa_schedule_cpu.py found a
    // step of two writes and no read in the emitted ISA — the epilogue re-uses these LDS buffers).
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        asm volatile("" ::"v"(fx[0][i]));
        asm volatile("" ::"v"(fx[1][i]));
    }
#pragma unroll
    for (int j = 0; j < WT; ++j) asm volatile("" ::"v"(fw[j]));
    if constexpr (DS_STAMP) {
        const long st_t1 = __builtin_amdgcn_s_memtime();
        if (p.slab && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(bz * gy + by) * gx + bx) * 32 + wave * 8;
            d[0] = st_t1 - st_t0; d[1] = st_p1 - st_k0; d[2] = st_p2 - st_k0; d[3] = nsteps;
            d[4] = st_r0 - st_k0; d[5] = __builtin_amdgcn_s_memrealtime() - st_r0; d[6] = st_k0;
        }
    }
    // ---- epilogue
    auto coord = [&](int i) {
        int row_l, col_l;
        tile_rc(i, row_l, col_l);
        ConvCoord c;
        if constexpr (PAIR) {                                           // rows 0 .. 15: sample b, rows 16 .. 31: sample b + 1 (pix counts through both)
            const int smp = row_l >> 4;
            c.ho = row_l & 15;
            c.wo = w0 + col_l;
            c.ok = c.ho < p.H && c.wo < p.W && (!smp || has2);
            c.pix = (smp * p.H + c.ho) * p.W + c.wo;
            return c;
        }
        c.ho = h0 + row_l;
        c.wo = w0 + col_l;
        c.ok = c.ho < p.H && c.wo < p.W;
        c.pix = c.ho * p.W + c.wo;
        return c;
    };
    auto coord2 = [&](int i, int mm) {                                  // pixel (tile i, lane mm): the contiguous side of halo3_epilogue_rows
        int row_l, col_l;
        if constexpr (TWL == 5) { row_l = 2 * wave + (i >> 1); col_l = 16 * (i & 1) + mm; }
        else if constexpr (TWL == 4) { row_l = 4 * wave + i; col_l = mm; }
        else { row_l = 8 * wave + i + 4 * (mm >> 3); col_l = mm & 7; }
        ConvCoord c;
        if constexpr (PAIR) {
            const int smp = row_l >> 4;
            c.ho = row_l & 15;
            c.wo = w0 + col_l;
            c.ok = c.ho < p.H && c.wo < p.W && (!smp || has2);
            c.pix = (smp * p.H + c.ho) * p.W + c.wo;
            return c;
        }
        c.ho = h0 + row_l;
        c.wo = w0 + col_l;
        c.ok = c.ho < p.H && c.wo < p.W;
        c.pix = c.ho * p.W + c.wo;
        return c;
    };
    char* const stage = smem + OFF_H + wave * EPI_F32_WAVE;              // (both halo buffers are dead: 4 x 6400 B of their 2 x HB)
    static_assert(4 * EPI_F32_WAVE <= 2 * G::HB, "the staging tiles fit the halo buffers");
    float s1 = 0.f, s2 = 0.f;
    const int outHW = p.H * p.W;
    const long st_e1 = DS_STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    const int out_mode = HP ? (p.flags >> 1) & 3 : 0;
    // PAIR: the second sample's shift table (its own mean) is built now, into the dead halo area behind the epilogue's staging tiles — the
    // fold-table entries are fetched again (L2: one more round trip per block, beside a K loop of 648 steps); waves 2, 3 then take that
    // table and that sample's GroupNorm factor.  The epilogues address `out` / `res` per PAIR of samples: pixel index 0 .. 2 H W.
    const float* shl_w = shl;
    float ga_w = gn_a;
    int b_e = b, outHW_e = outHW;
    if constexpr (PAIR) {
        // (its factor and mean come back from the LDS slot the prologue parked them in)
        const float gn_a1 = reinterpret_cast<const float*>(smem + G::LDS)[0], gn_am1 = reinterpret_cast<const float*>(smem + G::LDS)[1];
        const bool fold1 = fold;
        float* const shl1 = reinterpret_cast<float*>(smem + OFF_H + 4 * EPI_F32_WAVE);
        static_assert(4 * EPI_F32_WAVE + SHL_BYTES <= 2 * G::HB, "the second shift table fits the halo buffers too");
        const int ncls1 = ncls, cout1 = p.Cout;
#pragma unroll
        for (int k = 0; k < ST_IT; ++k) {
            const int e = tid + k * NT, cls = e / BN, n = n0 + e - cls * BN;
            float t1 = 0.f, t2 = 0.f;
            if (e < ncls1 * BN && n < cout1) {
                if (fold1) {
                    t1 = DS_LD(float, p.fold_t1 + cls * cout1 + n, DS_BX_T1);
                    t2 = DS_LD(float, p.fold_t2 + cls * cout1 + n, DS_BX_T2);
                } else if (p.bias) t1 = DS_LD(float, p.bias + n, DS_BX_BIAS);
            }
            if (e < 10 * BN) shl1[e] = t1 - gn_am1 * t2;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) >= 2) {      // (wave-uniform by construction: the table pointer and the factor stay scalar)
            shl_w = shl1;
            ga_w = gn_a1;
        }
        b_e = b >> 1;
        outHW_e = 2 * outHW;
    }
    if constexpr (HP) {
      if (raw) {                             // K slice of a split-precision launch: raw fp32 partial sums -> slab[kz][b][pixel][roundup(Cout, 8)]
        ds_conv_params q = p;
        q.out = p.slab;
        q.out_C = (p.Cout + 7) / 8 * 8;
        q.out_c0 = 0;
        q.gn_ab = nullptr;
        q.gn_part = nullptr;
        q.res = nullptr;
        halo3_epilogue_hp<DS_ACT_NONE, 2, false>(q, acc, kz * p.B + b, n0, outHW, shl, coord, s1, s2, 1.0f, lane);
      } else
      if (out_mode == 1) {                   // split bf16 planes (conv1 of a block in the split-precision tier: GELU, no residual)
        if (p.act == DS_ACT_GELU) halo3_epilogue_hp<DS_ACT_GELU, 1, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
        else halo3_epilogue_hp<DS_ACT_NONE, 1, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
      } else if (out_mode == 2) {          // fp32 (+ fp32 residual): conv2
        if (p.res && DS_EPI_ROWS) halo3_epilogue_rows_f32<true, true>(p, acc, b_e, n0, outHW_e, shl_w, coord, coord2, stage, s1, s2, ga_w, lane);
        else if (DS_EPI_ROWS) halo3_epilogue_rows_f32<true, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, coord2, stage, s1, s2, ga_w, lane);
        else if (p.res) halo3_epilogue_hp<DS_ACT_NONE, 2, true>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
        else halo3_epilogue_hp<DS_ACT_NONE, 2, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
      } else {                             // split input, bf16 output
        if (p.act == DS_ACT_GELU) halo3_epilogue<DS_ACT_GELU, true, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
        else halo3_epilogue<DS_ACT_NONE, true, false>(p, acc, b_e, n0, outHW_e, shl_w, coord, s1, s2, ga_w, lane);
      }
    } else
    if (raw) {                               // fp32 partial sums of this K slice -> slab[kz][b][pixel][roundup(Cout, 8)]
        ds_conv_params q = p;
        q.out = p.slab;
        q.out_C = (p.Cout + 7) / 8 * 8;
        q.out_c0 = 0;
        q.gn_ab = nullptr;
        q.gn_part = nullptr;
        halo3_epilogue_hp<DS_ACT_NONE, 2, false>(q, acc, kz * p.B + b, n0, outHW, shl, coord, s1, s2, 1.0f, lane);
    } else
    // (the border class costs a few selects per pixel tile: always computed; instantiations = activation x residual)
    if (p.act == DS_ACT_GELU) {
        const char* const lut = smem + G::OFF_LUT;
        if (use_lut) {
            if (p.res) halo3_epilogue<DS_ACT_GELU, true, true, true>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane, lut);
            else halo3_epilogue<DS_ACT_GELU, true, false, true>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane, lut);
        } else {
            if (p.res) halo3_epilogue<DS_ACT_GELU, true, true>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane);
            else halo3_epilogue<DS_ACT_GELU, true, false>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane);
        }
    } else {
        // with a residual the line-sized form wins 5-7 % on the 256x64 / 128x32 layers (its residual loads are whole lines too); without one the
        // register-only epilogue is 0.6 % faster (same-box A/B, profiles/r03_epilogue_rows_ab.txt)
        if (p.res && DS_EPI_ROWS) halo3_epilogue_rows<DS_ACT_NONE, true, true>(p, acc, b, n0, outHW, shl, coord, coord2, stage, s1, s2, gn_a, lane);
        else if (p.res) halo3_epilogue<DS_ACT_NONE, true, true>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane);
#if DS_EPI_ROWS_NORES
        else halo3_epilogue_rows<DS_ACT_NONE, true, false>(p, acc, b, n0, outHW, shl, coord, coord2, stage, s1, s2, gn_a, lane);
#else
        else halo3_epilogue<DS_ACT_NONE, true, false>(p, acc, b, n0, outHW, shl, coord, s1, s2, gn_a, lane);
#endif
    }
    long st_e2 = 0, st_e3 = 0;
    if constexpr (DS_STAMP) {
        st_e2 = __builtin_amdgcn_s_memrealtime();      // epilogue body issued (stores may still be in flight)
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): stores acknowledged
        st_e3 = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (p.stats_part && !raw) {
        const int parts = gridDim.x * gridDim.y;
        if constexpr (PAIR) {              // one partial per SAMPLE: waves 0, 1 -> sample b, waves 2, 3 -> sample b + 1
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {
                red[2 * wave] = s1;
                red[2 * wave + 1] = s2;
            }
            __syncthreads();
            if (lane == 0 && (wave == 0 || (wave == 2 && has2))) {
                float* const dst = p.stats_part + ((size_t)(b + (wave >> 1)) * parts + by * gx + bx) * 2;
                DS_ST(float, dst, DS_BX_STATS, red[2 * wave] + red[2 * wave + 2]);
                DS_ST(float, dst + 1, DS_BX_STATS, red[2 * wave + 1] + red[2 * wave + 3]);
            }
        } else
        block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * parts + by * gx + bx) * 2);
    }
    if constexpr (DS_STAMP) {
        if (p.slab && lane == 0) {
            long* d = reinterpret_cast<long*>(p.slab) + ((size_t)(bz * gy + by) * gx + bx) * 32 + wave * 8;
            d[7] = __builtin_amdgcn_s_memrealtime() - st_k0;
            if (DS_STAMP == 2) { d[1] = st_e2 - st_e1; d[2] = st_e3 - st_e2; }     // epilogue split instead of the prologue split
        }
    }
}

int halo3_twl(int W) {
    int twl = 3;
    while ((1 << twl) < W && twl < 5) ++twl;
    return twl;
}

}  // namespace

// statistics partials of a whole-K launch: one per block (pixel tile x N-block)
int ds_conv3x3_halo3_parts(const ds_conv_params* p) {
    const int twl = halo3_twl(p->W), TW = 1 << twl, TH = BM >> twl;
    return ((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW) * (p->cout_pad / BN);
}

int ds_conv3x3_halo3_launch(const ds_conv_params* p, hipStream_t st) {
    DS_REQUIRE(p->dtype == DS_BF16, "conv3x3_halo3: bf16 only");
    DS_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad_h == 1 && p->pad_w == 1 && !p->transposed,
               "conv3x3_halo3: 3x3 stride 1 pad 1 only");
    DS_REQUIRE(p->C1 == 0 && p->C0 % 32 == 0, "conv3x3_halo3: single source, Cin multiple of 32 (got %d+%d)", p->C0, p->C1);
    DS_REQUIRE(p->Ho == p->H && p->Wo == p->W && !p->out_nchw_f32, "conv3x3_halo3: same-size NHWC output only");
    DS_REQUIRE(p->cout_pad % BN == 0 && p->wk_order == 1, "conv3x3_halo3: cout_pad %% 96 == 0 and chunk-major weights (wk_order = 1)");
    const bool split_in = (p->flags & DS_CONV_F_SPLIT_IN) != 0;
    // (K slices of a split input are whole source chunks = triples of virtual chunks: C0 = 2C, C / 32 source chunks)
    DS_REQUIRE(p->ksplit <= 1 || (p->slab && (p->flags == 0 || split_in) && ((split_in ? p->C0 / 64 : p->C0 / 32) % p->ksplit) == 0 && !p->res_steps),
               "conv3x3_halo3: ksplit=%d needs a slab, must divide the %d source chunks and excludes the fused res_conv", p->ksplit, split_in ? p->C0 / 64 : p->C0 / 32);
    const int out_mode = (p->flags >> 1) & 3;
    DS_REQUIRE(out_mode <= 2 && (p->flags & ~7) == 0, "conv3x3_halo3: unknown flags %d", p->flags);
    DS_REQUIRE(!split_in || (p->C0 % 64 == 0 && !p->res_steps), "conv3x3_halo3: split input needs C0 = 2C with C %% 32 == 0 and no fused res_conv");
    DS_REQUIRE(p->flags == 0 || split_in, "conv3x3_halo3: the split-precision launches (flags != 0) take a split input (DS_CONV_F_SPLIT_IN)");
    DS_REQUIRE(out_mode == 0 || !p->res_steps, "conv3x3_halo3: split / fp32 output excludes the fused res_conv");
    DS_REQUIRE(out_mode != 2 || p->act == DS_ACT_NONE, "conv3x3_halo3: fp32 output has no activation variant");
    DS_REQUIRE(!p->res || out_mode == 2 || p->flags == 0, "conv3x3_halo3: in the split-precision modes a residual needs the fp32 output");
    DS_REQUIRE(out_mode != 1 || p->out_C >= p->out_c0 + 2 * p->Cout, "conv3x3_halo3: split output needs out_C >= 2 Cout");
    const int nchunks = split_in ? (p->C0 / 32) * 3 / 2 : p->C0 / 32;
    if (p->res_steps) {
        DS_REQUIRE(!p->res, "conv3x3_halo3: a fused res_conv excludes a residual tensor");
        DS_REQUIRE(p->res_src0 && p->res_C0 > 0 && p->res_C0 % 32 == 0 && p->res_C1 % 32 == 0 && p->res_steps == (p->res_C0 + p->res_C1) / 32 &&
                       p->res_steps % 3 == 0,
                   "conv3x3_halo3: res_conv channels (%d,%d) must be multiples of 32 (96 in total) and res_steps = their chunks", p->res_C0, p->res_C1);
        DS_REQUIRE(p->res_C1 == 0 || (p->res_src1 && p->res_H1 > 0 && p->res_W1 > 0), "conv3x3_halo3: second res_conv source incomplete");
        DS_REQUIRE(ds_aligned16(p->res_src0) && (!p->res_C1 || ds_aligned16(p->res_src1)), "conv3x3_halo3: res_conv sources must be 16-byte aligned");
    }
    DS_REQUIRE((long long)p->H * p->W * (p->C0 > p->res_C0 ? p->C0 : p->res_C0) * 2 < (1ll << 31) &&
                   (long long)(nchunks * 9 + p->res_steps) * p->cout_pad * 64 < (1ll << 31),
               "conv3x3_halo3: one sample / the packed weights must stay below 2 GiB (32-bit buffer offsets)");
    DS_REQUIRE((long long)p->H * p->W * p->out_C * (out_mode == 2 ? 4 : 2) < (1ll << 31), "conv3x3_halo3: one output sample must stay below 2 GiB (32-bit buffer offsets)");
    const int twl = halo3_twl(p->W), TW = 1 << twl, TH = BM >> twl;
    // r05: two samples per block where an image fills at most half of the 8 x 32 tile (the deepest level at 128 x 64 latents: 16 x 8)
    static const bool no_pair = getenv("DS_NO_HALO3_PAIR") != nullptr;                    // A/B switch
    const bool pair = !no_pair && p->flags != 0 && twl == 3 && 2 * p->H <= TH && p->W <= TW && p->ksplit <= 1 && !p->res_steps && p->B >= 2;
    dim3 grid(((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW), p->cout_pad / BN, pair ? (p->B + 1) / 2 : p->B * (p->ksplit > 1 ? p->ksplit : 1));
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_HALO);
        ds_conv_bounds_table(*p, DS_K_CONV_HALO, grid.x * grid.y, &h.t);
        h.set(DS_BX_W, p->wpk, (long long)(nchunks * 9 + p->res_steps) * p->cout_pad * 64);
        h.set(DS_BX_AUX0, p->res_steps ? p->res_src0 : nullptr, (long long)p->B * p->H * p->W * p->res_C0 * 2);
        h.set(DS_BX_AUX1, p->res_C1 ? p->res_src1 : nullptr, (long long)p->B * p->res_H1 * p->res_W1 * p->res_C1 * 2);
        h.set(DS_BX_AUX2, p->res_bias, (long long)p->Cout * 4);
        if (out_mode == 2) {                                  // fp32 output / residual: twice the bytes the bf16 description implies
            h.set(DS_BX_OUT, p->out, (long long)p->B * p->H * p->W * p->out_C * 4);
            h.set(DS_BX_RES, p->res, (long long)p->B * p->H * p->W * p->out_C * 4);
        }
        h.publish(st);
    }
#endif
    {
        static DsDevOnce once;                               // the GELU table of this device (module-scope __device__ array)
        int dev;
        if (once.need(&dev)) {
            float h[GELU_TAB_N];
            auto T = [](double a) { return a * 0.5 * erfc(a * 0.70710678118654752440); };
            for (int i = 0; i < GELU_TAB_N; ++i) {
                const unsigned lo = (unsigned)(GELU_TAB_BASE + i) << 16, hi = lo + 0x10000u;        // the bucket [lo, hi) of fp32 patterns
                float flo, fhi;
                memcpy(&flo, &lo, 4);
                memcpy(&fhi, &hi, 4);
                h[i] = (float)T(0.5 * ((double)flo + (double)fhi));
            }
            hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_gelu_lut), h, sizeof(h));
            if (e != hipSuccess) DS_FAIL(DS_ELAUNCH, "conv3x3_halo3: GELU table upload: %s", hipGetErrorString(e));
            once.done(dev);
        }
    }
    int lds = twl == 3 ? HG<3>::LDS : (twl == 4 ? HG<4>::LDS : HG<5>::LDS);
    static const bool no_lut = getenv("DS_HALO3_NOLUT") != nullptr;      // (A/B: the polynomial GELU)
#if DS_STAMP
    if (getenv("DS_HALO3_ONEBLOCK")) lds = 100 * 1024;      // diagnostic: one block per CU (lone-wave K loop timing)
#endif
#define DS_H3_LAUNCH(TWL_, HP_, PAIR_)                                                                        \
    do {                                                                                               \
        DS_SET_MAX_LDS((conv3x3_halo3_kernel<TWL_, HP_, PAIR_>), 100 * 1024, "conv3x3_halo3");                \
        hipLaunchKernelGGL((conv3x3_halo3_kernel<TWL_, HP_, PAIR_>), grid, dim3(NT), lds, st, *p, no_lut ? 0 : 1); \
    } while (0)
    if (pair) {
        lds += 64;                      // (the parked GroupNorm pair of the second sample)
        DS_H3_LAUNCH(3, true, true);
    }
    else if (p->flags) {
        if (twl == 5) DS_H3_LAUNCH(5, true, false);
        else if (twl == 4) DS_H3_LAUNCH(4, true, false);
        else DS_H3_LAUNCH(3, true, false);
    } else {
        if (twl == 5) DS_H3_LAUNCH(5, false, false);
        else if (twl == 4) DS_H3_LAUNCH(4, false, false);
        else DS_H3_LAUNCH(3, false, false);
    }
#undef DS_H3_LAUNCH
    DS_CHECK_LAUNCH("conv3x3_halo3");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv_halo3(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
