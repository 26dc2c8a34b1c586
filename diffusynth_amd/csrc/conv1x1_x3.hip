// 1x1 convolution of fp32 tensors in split precision on the bf16 matrix cores (tier "bf16x3"; gfx950).
//
// The U-Net's 1x1 convolutions — to_qkv (no bias, PreNorm folded), to_out, res_conv over pad_and_concat(encoder, decoder)
// (diffusion_components.py:128,139,263-264,210-249) — are 7 % of its MACs.  In the split-precision tier they were the last dense products
// on the fp32 kernel (conv_igemm<float>, fp32 MFMA = 1/16 of the bf16 rate): 35 ms of a 137 ms step.  Here
//     x . w  ~  x_hi . w_hi + x_lo . w_hi + x_hi . w_lo,   x_hi = bf16(x), x_lo = bf16(x - x_hi)   (the dropped x_lo . w_lo term is 2^-16 relative)
// with fp32 accumulation, like the 3x3 convolutions of that tier (conv3x3_halo3.hip, HP instantiation) — but the input tensor is read as
// fp32 and split in registers on its way to LDS (a 1x1 convolution reads every input element once per N-block: a separate
// ds_split_planes pass would cost as much traffic as the convolution itself).  Weights arrive pre-split: [chunk of 32 channels][hi, lo][cout_pad][32].
//
// Block tile 256 px x 96 channels, 4 waves of 64 px x 96 ch on v_mfma_f32_16x16x32_bf16, accumulator layout, LDS swizzles and the fp32
// epilogue (GroupNorm fold, bias, optional fp32 residual, statistics partials) shared with conv3x3_halo3 (conv_halo3_common.hpp).
// The kernel is bound by its fp32 traffic (to_qkv at 256 x 64, batch 128: 0.8 GB in, 3.2 GB out), not by the matrix pipe: a plain
// two-barrier K loop with a register prefetch of the next chunk, two blocks per CU.
#include "common.hpp"
#include "conv_halo3_common.hpp"

namespace {

constexpr int X3_OFF_XH = 0, X3_OFF_XL = BM * PSTR, X3_OFF_WH = 2 * BM * PSTR, X3_OFF_WL = X3_OFF_WH + B_BYTES, X3_OFF_SHL = X3_OFF_WL + B_BYTES;
constexpr int X3_OFF_RED = X3_OFF_SHL + SHL_BYTES, X3_LDS = X3_OFF_RED + 64;      // 32768 + 12288 + 3840 + 64 = 48960
constexpr int X3_XIT = BM * 8 / NT;                 // 16-byte fp32 pieces of a 256 px x 32 ch chunk per thread: 8
constexpr int X3_WIT = 2 * BN * 4 / NT;             // 16-byte pieces of the two weight planes per thread: 3

static_assert(4 * EPI_F32_WAVE <= X3_OFF_SHL, "the epilogue's staging tiles reuse the x / w buffers, the shift table stays");

__global__ __launch_bounds__(NT, DS_MINBLK) void conv1x1_x3_kernel(const ds_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const shl = reinterpret_cast<float*>(smem + X3_OFF_SHL);
    float* const red = reinterpret_cast<float*>(smem + X3_OFF_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    // XCD-chunked block order, N-block fastest (as conv3x3_halo3): the N-blocks of one pixel tile run back to back on ONE XCD, so the fp32 input
    // tile comes from HBM once and from that XCD's L2 for the other N-blocks (to_qkv has four)
    const int gy = p.cout_pad / BN, HW = p.H * p.W, gx = (HW + BM - 1) / BM, nwg = gridDim.x;
    int wid = blockIdx.x;
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
    const int by = wid % gy, bxz = wid / gy, bx = bxz % gx;
    // split-K (r04, small batches: a 64 x 16-level res_conv at batch 1 was 16 blocks of 24 - 36 serial chunks, 30 us): grid z = K slice * B +
    // sample; a slice runs chunks [c_lo, c_hi) and stores raw fp32 partial sums to p.slab[zb], ds_conv_splitk_reduce finishes
    const int zb = bxz / gx, ksplit = p.ksplit > 1 ? p.ksplit : 1, kz = zb / p.B;
    const int b = zb - kz * p.B, n0 = by * BN, px0 = bx * BM;
    const int NC0 = p.C0 >> 5, NCC = (p.C0 + p.C1) >> 5;
    const int ncs = (NCC + ksplit - 1) / ksplit, c_lo = kz * ncs, c_hi = min(NCC, c_lo + ncs);
    const bool raw = p.ksplit > 1;

    // ---- loads of a chunk: piece (it, tid) = pixel (tid >> 3) + 32 it, channels 4 (tid & 7) .. + 3 of the chunk (8 lanes = one pixel's 128 bytes)
    const float* const sp0 = reinterpret_cast<const float*>(p.src0) + (size_t)b * HW * p.C0;
    const float* const sp1 = p.C1 ? reinterpret_cast<const float*>(p.src1) + (size_t)b * p.H1 * p.W1 * p.C1 : nullptr;
    // byte offset of the piece's pixel in each source; bit 31 set = zero (outside the image / the padded decoder map): the loads are range-checked
    // buffer loads, so "outside" costs no select on the loaded data (a select makes the wave wait for the load where it is issued, and the
    // prefetch of the next chunk then no longer overlaps the MFMAs of this one)
    unsigned off0[X3_XIT], off1[X3_XIT];
#pragma unroll
    for (int it = 0; it < X3_XIT; ++it) {
        const int px = px0 + (tid >> 3) + 32 * it;
        off0[it] = ((unsigned)(px * p.C0 + 4 * (tid & 7)) * 4u & 0x7fffffffu) | ((unsigned)(px >= HW) << 31);
        off1[it] = 0x80000000u;
        if (p.C1) {
            const int h = px / p.W - p.off_h1, w = px % p.W - p.off_w1;       // pad_and_concat: the second source sits at (off_h1, off_w1)
            const unsigned bad = (unsigned)(px >= HW) | (unsigned)((unsigned)h >= (unsigned)p.H1) | (unsigned)((unsigned)w >= (unsigned)p.W1);
            off1[it] = ((unsigned)((h * p.W1 + w) * p.C1 + 4 * (tid & 7)) * 4u & 0x7fffffffu) | (bad << 31);
        }
    }
    const rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp0), (short)0, HW * p.C0 * 4, 0x00020000);
    const rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.C1 ? sp1 : sp0), (short)0, p.C1 ? p.H1 * p.W1 * p.C1 * 4 : 0, 0x00020000);
    const char* const wbase = reinterpret_cast<const char*>(p.wpk);
    const size_t wchunk = (size_t)2 * p.cout_pad * 64;                          // bytes of one chunk: [hi, lo][cout_pad][32] bf16
    f32x4 rx[X3_XIT];
    u32x4 rw[X3_WIT];
    auto load_chunk = [&](int cc) {
        const bool first = cc < NC0;
        const float* const s = first ? sp0 : sp1;
        const int c0 = (first ? cc : cc - NC0) * 32;
#if DS_BOUNDS
#pragma unroll
        for (int it = 0; it < X3_XIT; ++it) {
            const unsigned o = first ? off0[it] : off1[it];
            const bool in = o < 0x80000000u;
            const f32x4 v = DS_LD(f32x4, reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s) + (in ? o + c0 * 4 : 0)), first ? DS_BX_SRC0 : DS_BX_SRC1);
            rx[it] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#else
        (void)s;
#pragma unroll
        for (int it = 0; it < X3_XIT; ++it)
            rx[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(first ? rs0 : rs1, (int)((first ? off0[it] : off1[it]) + c0 * 4), 0, 0));
#endif
#pragma unroll
        for (int it = 0; it < X3_WIT; ++it) {
            const int piece = tid + it * NT, plane = piece / (BN * 4), r = piece - plane * (BN * 4);      // r = row * 4 + quarter
            rw[it] = DS_LD(u32x4, reinterpret_cast<const u32x4*>(wbase + (size_t)cc * wchunk + ((size_t)plane * p.cout_pad + n0) * 64 + (size_t)r * 16), DS_BX_W);
        }
    };
    // LDS images: the 64-byte rows and swizzles of conv3x3_halo3 (pixel rows: quarter ^ 2 * bit2(pixel); weight rows: quarter ^ W_SWZ(row))
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < X3_XIT; ++it) {
            const int pl = (tid >> 3) + 32 * it, e = tid & 7;                   // 8-byte slot e of the pixel's row: quarter e >> 1, half e & 1
            const int a = pl * PSTR + ((((e >> 1) ^ (((pl >> 2) & 1) << 1))) << 4) + ((e & 1) << 3);
            const f32x4 v = rx[it];
            uint2 hi, lo;
            ds_split2(v[0], v[1], hi.x, lo.x);
            ds_split2(v[2], v[3], hi.y, lo.y);
            *reinterpret_cast<uint2*>(smem + X3_OFF_XH + a) = hi;
            *reinterpret_cast<uint2*>(smem + X3_OFF_XL + a) = lo;
        }
#pragma unroll
        for (int it = 0; it < X3_WIT; ++it) {
            const int piece = tid + it * NT, plane = piece / (BN * 4), r = piece - plane * (BN * 4), row = r >> 2, qq = r & 3;
            *reinterpret_cast<u32x4*>(smem + (plane ? X3_OFF_WL : X3_OFF_WH) + row * PSTR + ((qq ^ W_SWZ(row)) << 4)) = rw[it];
        }
    };

    load_chunk(c_lo);
    // ---- GroupNorm fold / bias -> shift table (row 0; the epilogue's row 9 is the zero row of lanes without a pixel)
    float gn_a = 1.f, gn_am = 0.f;
    const bool fold = p.gn_ab != nullptr && !raw;                      // (a K slice: zero shift, factor 1 — the reduce applies fold / bias)
    if (fold) {
        gn_a = DS_LD(float, p.gn_ab + 2 * b, DS_BX_GNAB);
        gn_am = DS_LD(float, p.gn_ab + 2 * b + 1, DS_BX_GNAB);
    }
    for (int e = tid; e < 10 * BN; e += NT) {
        float v = 0.f;
        const int n = n0 + e;
        if (e < BN && n < p.Cout) {
            if (fold) v = DS_LD(float, p.fold_t1 + n, DS_BX_T1) - gn_am * DS_LD(float, p.fold_t2 + n, DS_BX_T2);
            else if (p.bias && !raw) v = DS_LD(float, p.bias + n, DS_BX_BIAS);
        }
        shl[e] = v;
    }
    // fragment addresses: pixel (tile i, lane m) = 64 wave + 16 i + m; weight rows as in conv3x3_halo3
    int xa[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        const int pl = 64 * wave + 16 * i + m;
        xa[i] = pl * PSTR + ((q ^ (((pl >> 2) & 1) << 1)) << 4);
    }
    const int wa = W_ROW0(m) * PSTR + ((q ^ ((-(m >> 2)) & 3)) << 4);                      // + EPI_CH(j) rows for tile j
    f32x4 acc[XT][WT];
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int cc = c_lo; cc < c_hi; ++cc) {
        store_chunk();
        __syncthreads();
        if (cc + 1 < c_hi) load_chunk(cc + 1);
        bf16x8 xh[XT], xl[XT];
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            xh[i] = *reinterpret_cast<const bf16x8*>(smem + X3_OFF_XH + xa[i]);
            xl[i] = *reinterpret_cast<const bf16x8*>(smem + X3_OFF_XL + xa[i]);
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(smem + X3_OFF_WH + wa + EPI_CH(j) * PSTR);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(smem + X3_OFF_WL + wa + EPI_CH(j) * PSTR);
#pragma unroll
            for (int i = 0; i < XT; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[i], acc[i][j], 0, 0, 0);     // small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[i], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // fp32 epilogue with line-sized stores (halo3_epilogue_rows_f32): pixel (tile i, lane mm) of this wave = px0 + 64 wave + 16 i + mm
    auto coord2 = [&](int i, int mm) {
        ConvCoord c;
        const int px = px0 + 64 * wave + 16 * i + mm;
        c.ok = px < HW;
        c.ho = 0;
        c.wo = 0;
        c.pix = px;
        return c;
    };
    auto coord = [&](int i) { return coord2(i, m); };
    float s1 = 0.f, s2 = 0.f;
    char* const stage = smem + wave * EPI_F32_WAVE;                     // (the K loop ended with a barrier: the x / w images are dead)
    if (raw) {                                                          // slab[kz][b][pixel][roundup(Cout, 8)] through the same fp32 epilogue
        ds_conv_params qp = p;
        qp.out = p.slab;
        qp.out_C = (p.Cout + 7) / 8 * 8;
        qp.out_c0 = 0;
        qp.res = nullptr;
        halo3_epilogue_rows_f32<false, false>(qp, acc, zb, n0, HW, shl, coord, coord2, stage, s1, s2, 1.0f, lane);
        return;
    }
    if (p.res) halo3_epilogue_rows_f32<false, true>(p, acc, b, n0, HW, shl, coord, coord2, stage, s1, s2, gn_a, lane);
    else halo3_epilogue_rows_f32<false, false>(p, acc, b, n0, HW, shl, coord, coord2, stage, s1, s2, gn_a, lane);
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * (gx * gy) + by * gx + bx) * 2);
}

}  // namespace

extern "C" size_t ds_conv1x1_x3_weight_elems(int Cin, int Cout) { return (size_t)((Cin + 31) / 32) * 2 * ((Cout + BN - 1) / BN * BN) * 32; }

int ds_conv_splitk_parts(const ds_conv_params* p);                     // conv_splitk.hip
extern "C" int ds_conv1x1_x3_stats_parts(const ds_conv_params* p) {
    return p->ksplit > 1 ? ds_conv_splitk_parts(p) : ((p->H * p->W + BM - 1) / BM) * (p->cout_pad / BN);
}

extern "C" int ds_conv1x1_x3(const ds_conv_params* p, void* stream) {
    DS_REQUIRE(p && p->src0 && p->wpk && p->out, "conv1x1_x3: null pointer");
    DS_REQUIRE(p->KH == 1 && p->KW == 1 && p->stride == 1 && p->pad_h == 0 && p->pad_w == 0 && !p->transposed && p->Ho == p->H && p->Wo == p->W,
               "conv1x1_x3: 1x1 stride 1 only");
    DS_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->Cout > 0, "conv1x1_x3: empty problem");
    DS_REQUIRE(p->C0 > 0 && p->C0 % 32 == 0 && p->C1 >= 0 && p->C1 % 32 == 0, "conv1x1_x3: channel counts (%d,%d) must be multiples of 32", p->C0, p->C1);
    DS_REQUIRE(p->C1 == 0 || (p->src1 && p->H1 > 0 && p->W1 > 0), "conv1x1_x3: second source incomplete");
    DS_REQUIRE(p->cout_pad % BN == 0 && p->Cout <= p->cout_pad, "conv1x1_x3: cout_pad %d must be a multiple of %d", p->cout_pad, BN);
    DS_REQUIRE(p->flags == (DS_CONV_F_IN_F32 | DS_CONV_F_OUT_F32) && p->act == DS_ACT_NONE && !p->res_steps && !p->gn_part && !p->out_nchw_f32,
               "conv1x1_x3: fp32 in / fp32 out (flags = DS_CONV_F_IN_F32 | DS_CONV_F_OUT_F32), no activation, no fused res_conv, statistics through gn_ab");
    {
        const int nq = (p->C0 + p->C1) / 32, ks = p->ksplit, nqs = ks > 1 ? (nq + ks - 1) / ks : nq;
        DS_REQUIRE(ks <= 1 || (p->slab && (ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8) && (ks - 1) * nqs < nq),
                   "conv1x1_x3: ksplit=%d needs a slab, a value in {2, 3, 4, 6, 8} and no empty slice over %d chunks", ks, nq);
    }
    DS_REQUIRE(!p->gn_ab || (p->fold_t1 && p->fold_t2 && p->ncls == 1), "conv1x1_x3: the GroupNorm fold needs t1 / t2 tables with one border class");
    DS_REQUIRE(p->out_C % 4 == 0 && p->out_c0 % 4 == 0 && p->out_C >= p->out_c0 + p->Cout && p->Cout % 8 == 0, "conv1x1_x3: out_C / out_c0 multiples of 4, Cout a multiple of 8");
    DS_REQUIRE((long long)p->H * p->W * p->out_C * 4 < (1ll << 31) && (long long)p->H * p->W * p->C0 * 4 < (1ll << 31) && (long long)p->H1 * p->W1 * p->C1 * 4 < (1ll << 31),
               "conv1x1_x3: one sample of the output and of either source must stay below 2 GiB (32-bit buffer offsets)");
    if (!ds_aligned16(p->src0) || !ds_aligned16(p->wpk) || !ds_aligned16(p->out) || (p->C1 && !ds_aligned16(p->src1)) || (p->res && !ds_aligned16(p->res)))
        DS_FAIL(DS_EALIGN, "conv1x1_x3: tensors must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int gx = (p->H * p->W + BM - 1) / BM, gy = p->cout_pad / BN;
    dim3 grid(gx * gy * p->B * (p->ksplit > 1 ? p->ksplit : 1));
#if DS_BOUNDS
    {
        DsBxHost h(DS_K_CONV_IGEMM);
        const int NCC = (p->C0 + p->C1) / 32;
        h.set(DS_BX_SRC0, p->src0, (long long)p->B * p->H * p->W * p->C0 * 4);
        h.set(DS_BX_SRC1, p->C1 ? p->src1 : nullptr, (long long)p->B * p->H1 * p->W1 * p->C1 * 4);
        h.set(DS_BX_W, p->wpk, (long long)NCC * 2 * p->cout_pad * 64);
        if (p->ksplit > 1) h.set(DS_BX_OUT, p->slab, (long long)p->ksplit * p->B * p->H * p->W * ((p->Cout + 7) / 8 * 8) * 4);
        else h.set(DS_BX_OUT, p->out, (long long)p->B * p->H * p->W * p->out_C * 4);
        h.set(DS_BX_RES, p->res, (long long)p->B * p->H * p->W * p->out_C * 4);
        h.set(DS_BX_BIAS, p->bias, (long long)p->Cout * 4);
        h.set(DS_BX_T1, p->fold_t1, (long long)p->Cout * 4).set(DS_BX_T2, p->fold_t2, (long long)p->Cout * 4);
        h.set(DS_BX_GNAB, p->gn_ab, (long long)p->B * 2 * 4);
        h.set(DS_BX_STATS, p->stats_part, (long long)p->B * gx * gy * 2 * 4);
        h.publish(st);
    }
#endif
    DS_SET_MAX_LDS(conv1x1_x3_kernel, X3_LDS, "conv1x1_x3");
    hipLaunchKernelGGL(conv1x1_x3_kernel, grid, dim3(NT), X3_LDS, st, *p);
    DS_CHECK_LAUNCH("conv1x1_x3");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_conv1x1_x3(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif
