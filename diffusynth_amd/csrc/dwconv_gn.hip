// Depthwise 7x7 (+bias +time bias, two-source skip concat) and the GroupNorm kernels (gfx950).
// All of these are HBM-bound: 16-byte vector accesses along the channels-last C axis, fp32 math.
#include <type_traits>

#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ dwconv7
// thread = (channel vector, w, strip of TH rows); a 7-wide column of TH+6 inputs is held in registers per
// horizontal tap so every loaded value feeds up to TH outputs.
constexpr int DW_TH = 4;
constexpr int DW_BLOCK = 256;

template <typename T>
__global__ __launch_bounds__(DW_BLOCK) void dwconv7_kernel(const ds_dwconv_params p, int nstrip, int CV) {
    constexpr int V = Vec16<T>::N;
    __shared__ float red[2 * (DW_BLOCK / 64)];
    const int C = p.C0 + p.C1;
    const int b = blockIdx.y;
    const long gid = (long)blockIdx.x * DW_BLOCK + threadIdx.x;
    const long total = (long)nstrip * p.W * CV;
    float s1 = 0.f, s2 = 0.f;
    if (gid < total) {
        const int cv = gid % CV;
        const int w = (gid / CV) % p.W;
        const int strip = gid / ((long)CV * p.W);
        const int h0 = strip * DW_TH;
        const int c = cv * V;
        const T* base;
        int Cs, cc, Hs, Ws, oh, ow;
        if (c < p.C0) {
            base = reinterpret_cast<const T*>(p.src0) + (size_t)b * p.H * p.W * p.C0;
            Cs = p.C0; cc = c; Hs = p.H; Ws = p.W; oh = 0; ow = 0;
        } else {
            base = reinterpret_cast<const T*>(p.src1) + (size_t)b * p.H1 * p.W1 * p.C1;
            Cs = p.C1; cc = c - p.C0; Hs = p.H1; Ws = p.W1; oh = p.off_h1; ow = p.off_w1;
        }
        float acc[DW_TH][V];
        float init[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            init[v] = p.bias[c + v];
            if (p.tbias) init[v] += p.tbias[(size_t)b * p.tb_stride + c + v];
        }
#pragma unroll
        for (int t = 0; t < DW_TH; ++t)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[t][v] = init[v];

        for (int dw = 0; dw < 7; ++dw) {
            const int wi = w + dw - 3 - ow;
            if ((unsigned)wi >= (unsigned)Ws) continue;  // zero column
            float col[DW_TH + 6][V];
#pragma unroll
            for (int r = 0; r < DW_TH + 6; ++r) {
                const int hi = h0 + r - 3 - oh;
                if ((unsigned)hi < (unsigned)Hs) {
                    vec16_load<T>(base + ((size_t)(hi * Ws + wi) * Cs + cc), col[r], c < p.C0 ? DS_BX_SRC0 : DS_BX_SRC1);
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) col[r][v] = 0.f;
                }
            }
#pragma unroll
            for (int dh = 0; dh < 7; ++dh) {
                float wv[V];
                const float* wp = p.wt + (size_t)(dh * 7 + dw) * C + c;
#pragma unroll
                for (int v = 0; v < V; v += 4) {
                    f32x4 t4 = DS_LD(f32x4, wp + v, DS_BX_W);
                    wv[v] = t4[0]; wv[v + 1] = t4[1]; wv[v + 2] = t4[2]; wv[v + 3] = t4[3];
                }
#pragma unroll
                for (int t = 0; t < DW_TH; ++t)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[t][v] = fmaf(col[t + dh][v], wv[v], acc[t][v]);
            }
        }
        T* outp = reinterpret_cast<T*>(p.out) + (size_t)b * p.H * p.W * C;
#pragma unroll
        for (int t = 0; t < DW_TH; ++t) {
            const int h = h0 + t;
            if (h < p.H) {
                vec16_store<T>(outp + ((size_t)(h * p.W + w) * C + c), acc[t], DS_BX_OUT);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    s1 += acc[t][v];
                    s2 += acc[t][v] * acc[t][v];
                }
            }
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + blockIdx.x) * 2);
}

// ------------------------------------------------------------------------------------------------ dwconv7, LDS-tiled
// Block = 16 x 32 output pixels x CB channels (CB = 4 x 16-byte vectors: 32 bf16 / 16 fp32 channels).
// The 22 x 38 input halo is staged once in LDS (64 B per pixel, lanes -> consecutive 16-B chunks =>
// conflict-free ds_read_b128), the 49 x CB weights next to it.  Each thread owns one channel vector of an
// 8-row output strip: per horizontal tap it walks 14 input rows, every value feeding up to 7 outputs.
// r03: the tile is 512 pixels in three shapes — 32 x 16, 16 x 32 for images at most 16 wide and 8 x 64 for images at most 8 wide (a 32-wide tile
// spent half / three quarters of its threads on columns outside a 64x16 / 32x8 image: 1.0-1.7 TB/s there against 2.4 on the wide levels).
// NV = 16-byte channel vectors per pixel held in LDS per block.  4 (32 bf16 / 16 fp32 channels, a 512-pixel tile) was the only form until the
// end of r03; fp32 tensors whose channel counts allow it now take NV = 8 (32 channels = one whole 128-byte line per pixel, a 256-pixel tile):
// with 16 channels the kernel moved 64-byte load pieces and 32-byte store pieces and sat at the L2's REQUEST rate (C = 192 at 256x64: 182 M
// requests in 1360 us = 134 G/s with six of its seven tap columns removed, i.e. without its arithmetic) — not at a byte rate.  2 measured slower.
#ifndef DS_DW_SR
// output rows per thread of the LDS-tile kernel.  r04: 8 (256 threads per block) — per tap column a thread reads 7 weight + 14 input vectors for
// 8 x 7 outputs instead of 7 + 10 for 4 x 7 (38 % fewer LDS reads per output; the kernel is LDS-bound where it is not HBM-bound): 4 - 6 % per
// layer at the split-precision tier's shapes (same box, tools/ab.sh -m dw -a "--dtype fp32split"); 16 rows (128 threads: too few waves) is 12 - 25 % slower than 4
#define DS_DW_SR 8
#endif
constexpr int LT_SR = DS_DW_SR;
constexpr int LT_NT = 2048 / LT_SR;                             // threads: channel vectors x tile columns x row strips = NV x (2048 / NV pixels) / LT_SR
template <int TWL, int NV>
struct LT {
    static constexpr int W = 1 << TWL, H = (2048 / NV) >> TWL, HC = W + 6, HR = H + 6, NPX = HC * HR;
};
static int lt_twl(int W, int nv) { return W <= 8 ? 3 : ((W <= 16 || nv == 8) ? 4 : 5); }     // NV = 8: 16 x 16 (8 x 32 for images at most 8 wide)

template <typename T, int TWL, int LT_NV>
__global__ __launch_bounds__(LT_NT) void dwconv7_lds_kernel(const ds_dwconv_params p, int tiles_w, int tiles_hw, int ncblk) {
    constexpr int LT_W = LT<TWL, LT_NV>::W, LT_H = LT<TWL, LT_NV>::H, LT_HC = LT<TWL, LT_NV>::HC, LT_NPX = LT<TWL, LT_NV>::NPX;
    static_assert(LT_NV * LT_W * (LT_H / LT_SR) == LT_NT && LT_H % LT_SR == 0, "thread map");
    constexpr int V = Vec16<T>::N;
    constexpr int CB = LT_NV * V;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    uint4* xs = reinterpret_cast<uint4*>(dsm);                                          // [LT_NPX][LT_NV]
    float* wsm = reinterpret_cast<float*>(dsm + (size_t)LT_NPX * LT_NV * 16);          // [49][CB]
    float* red = wsm + 49 * CB;
    const int tid = threadIdx.x;
    // Block order: channel block fastest, then the tile, then the sample (plain hardware order).  r04 tried the XCD-chunked order of the 3x3
    // kernels here (-DDS_DW_XCD: the tiles of one (sample, channel block) plane consecutive on ONE XCD, so that the 6 halo rows / columns a
    // tile shares with its neighbours come from that XCD's L2 — the kernel fetches 1.47 x its input from HBM in plain order, PMC FETCH_SIZE):
    // 456 -> 483 us average on the fp32 levels, same box.  A plane's tiles then stream from the same few HBM channels at once; the plain
    // order spreads every moment's requests over all of them, and this kernel is bound by bytes in flight, not by bytes.
    const int gx0 = gridDim.x, nwg = gx0 * gridDim.y;
    int wid = blockIdx.x + gx0 * blockIdx.y;
#ifdef DS_DW_XCD
    if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
    const int tile = wid % tiles_hw, cblk = (wid / tiles_hw) % ncblk, b = wid / (tiles_hw * ncblk);
#else
    (void)nwg;
    const int cblk = wid % ncblk, tile = (wid / ncblk) % tiles_hw, b = wid / (tiles_hw * ncblk);
#endif
    const int bix = tile * ncblk + cblk;                      // (the block's slot in its sample's statistics partials: order-independent sum)
    const int th = tile / tiles_w, tw = tile - th * tiles_w;
    const int h0 = th * LT_H, w0 = tw * LT_W, c0 = cblk * CB;
    const int C = p.C0 + p.C1;
    const T* base;
    int Cs, cc, Hs, Ws, oh, ow;
    if (c0 < p.C0) {
        base = reinterpret_cast<const T*>(p.src0) + (size_t)b * p.H * p.W * p.C0;
        Cs = p.C0; cc = c0; Hs = p.H; Ws = p.W; oh = 0; ow = 0;
    } else {
        base = reinterpret_cast<const T*>(p.src1) + (size_t)b * p.H1 * p.W1 * p.C1;
        Cs = p.C1; cc = c0 - p.C0; Hs = p.H1; Ws = p.W1; oh = p.off_h1; ow = p.off_w1;
    }
#if DS_BOUNDS
    for (int slot = tid; slot < LT_NPX * LT_NV; slot += LT_NT) {
        const int px = slot / LT_NV, v = slot - px * LT_NV;
        const int hr = px / LT_HC, hc = px - hr * LT_HC;
        const int hi = h0 + hr - 3 - oh, wi = w0 + hc - 3 - ow;
        uint4 val = make_uint4(0, 0, 0, 0);
        if ((unsigned)hi < (unsigned)Hs && (unsigned)wi < (unsigned)Ws)
            val = DS_LD(uint4, base + ((size_t)(hi * Ws + wi) * Cs + cc + v * V), c0 < p.C0 ? DS_BX_SRC0 : DS_BX_SRC1);
        xs[slot] = val;
    }
    for (int i = tid; i < 49 * CB; i += LT_NT) wsm[i] = DS_LD(float, p.wt + (size_t)(i / CB) * C + c0 + (i % CB), DS_BX_W);
#else
    {
        // every halo piece of this thread is requested before the first one is written to LDS: range-checked buffer loads with arithmetic
        // out-of-range offsets (bit 31 set = beyond any sample; the launcher keeps samples below 2 GB) — `if (inside) v = load` is an
        // exec-masked region per iteration that waits for its own load before the next one is issued (seven serial round trips)
        constexpr int SLOTS = LT_NPX * LT_NV, ITS = (SLOTS + LT_NT - 1) / LT_NT;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), (short)0, (int)((size_t)Hs * Ws * Cs * sizeof(T)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), (short)0, 49 * C * 4, 0x00020000);
        u32x4 hv[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int slot = tid + it * LT_NT;
            const int px = slot / LT_NV, v = slot - px * LT_NV;
            const int hr = px / LT_HC, hc = px - hr * LT_HC;
            const int hi = h0 + hr - 3 - oh, wi = w0 + hc - 3 - ow;
            const unsigned bad = (unsigned)(slot >= SLOTS) | (unsigned)((unsigned)hi >= (unsigned)Hs) | (unsigned)((unsigned)wi >= (unsigned)Ws);
            const unsigned off = (unsigned)((hi * Ws + wi) * Cs + cc + v * V) * (unsigned)sizeof(T);
            hv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((off & 0x7fffffffu) | (bad << 31)), 0, 0);
        }
        // the block's 49 x CB weights: one 16-byte piece per thread (CB / 4 pieces per tap)
        constexpr int WP = 49 * CB / 4, WIT = (WP + LT_NT - 1) / LT_NT;
        asm volatile("" : "+v"(hv[ITS - 1]));                       // (the last, conditional piece: keep its load with the others)
        u32x4 wv4[WIT];
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int wp = tid + k * LT_NT, wtap = wp / (CB / 4), wj = wp - wtap * (CB / 4);
            wv4[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((unsigned)((wtap * C + c0 + 4 * wj) * 4) | ((unsigned)(wp >= WP) << 31)), 0, 0);
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int slot = tid + it * LT_NT;
            if (it + 1 < ITS || slot < SLOTS) *reinterpret_cast<u32x4*>(xs + slot) = hv[it];
        }
#pragma unroll
        for (int k = 0; k < WIT; ++k)
            if (tid + k * LT_NT < WP) *reinterpret_cast<u32x4*>(wsm + 4 * (tid + k * LT_NT)) = wv4[k];
    }
#endif
    const int cv = tid % LT_NV, wl = (tid / LT_NV) % LT_W, strip = tid / (LT_NV * LT_W);
    const int c = c0 + cv * V;
    float acc[LT_SR][V];
    {
        float init[V];
#if DS_BOUNDS
#pragma unroll
        for (int v = 0; v < V; ++v) {
            init[v] = p.bias[c + v];
            if (p.tbias) init[v] += p.tbias[(size_t)b * p.tb_stride + c + v];
        }
#else
        // bias and time bias as 16-byte buffer loads requested before the barrier (a NULL time bias is a zero-length buffer: loads return 0)
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), (short)0, C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.tbias ? p.tbias + (size_t)b * p.tb_stride : p.bias),
                                                                               (short)0, p.tbias ? C * 4 : 0, 0x00020000);
#pragma unroll
        for (int v = 0; v < V; v += 4) {
            const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (c + v) * 4, 0, 0), t4 = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (c + v) * 4, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) init[v + j] = __uint_as_float(b4[j]) + __uint_as_float(t4[j]);
        }
#endif
        __syncthreads();
#pragma unroll
        for (int o = 0; o < LT_SR; ++o)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[o][v] = init[v];
    }
#pragma unroll 1
    for (int dw = 0; dw < 7; ++dw) {   // not unrolled: keeps only one tap column of weights + inputs live
        float wv[7][V];
#pragma unroll
        for (int dh = 0; dh < 7; ++dh)
#pragma unroll
            for (int v = 0; v < V; v += 4) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(wsm + (dh * 7 + dw) * CB + cv * V + v);
                wv[dh][v] = t4[0]; wv[dh][v + 1] = t4[1]; wv[dh][v + 2] = t4[2]; wv[dh][v + 3] = t4[3];
            }
#pragma unroll
        for (int r = 0; r < LT_SR + 6; ++r) {
            float x[V];
            Vec16<T>::load(reinterpret_cast<const T*>(xs + ((strip * LT_SR + r) * LT_HC + wl + dw) * LT_NV + cv), x);
#pragma unroll
            for (int dh = 0; dh < 7; ++dh) {
                const int o = r - dh;
                if (o >= 0 && o < LT_SR) {
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[o][v] = fmaf(x[v], wv[dh][v], acc[o][v]);
                }
            }
        }
    }
    float s1 = 0.f, s2 = 0.f;
    T* outp = reinterpret_cast<T*>(p.out) + (size_t)b * p.H * p.W * C;
    const int w = w0 + wl;
#pragma unroll
    for (int o = 0; o < LT_SR; ++o) {
        const int h = h0 + strip * LT_SR + o;
        if (h < p.H && w < p.W) {
            if constexpr (sizeof(T) == 4) {
                if (p.out_split) {
                    // split-precision tier: the fp32 result as two bf16 planes (hi, then lo = v - hi) of a 2C-channel image,
                    // the input format of the split 3x3 convolution that follows (DS_CONV_F_SPLIT_IN)
                    bf16* o2 = reinterpret_cast<bf16*>(p.out) + ((size_t)b * p.H * p.W + (size_t)(h * p.W + w)) * (2 * C) + c;
                    uint2 hi, lo;
                    ds_split2(acc[o][0], acc[o][1], hi.x, lo.x);
                    ds_split2(acc[o][2], acc[o][3], hi.y, lo.y);
                    DS_ST(bf16x4, o2, DS_BX_OUT, __builtin_bit_cast(bf16x4, hi));
                    DS_ST(bf16x4, o2 + C, DS_BX_OUT, __builtin_bit_cast(bf16x4, lo));
                } else vec16_store<T>(outp + ((size_t)(h * p.W + w) * C + c), acc[o], DS_BX_OUT);
            } else vec16_store<T>(outp + ((size_t)(h * p.W + w) * C + c), acc[o], DS_BX_OUT);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                s1 += acc[o][v];
                s2 += acc[o][v] * acc[o][v];
            }
        }
    }
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * gridDim.x + bix) * 2);
}

// ------------------------------------------------------------------------------------------------ dwconv7, LDS ring over a column strip (r05)
// The tile kernel above fetches a 22 x 22 halo for every 16 x 16 outputs: 1.89 x its input from L2 and — PMC FETCH_SIZE, r04 — 1.46 x from
// HBM (the 6 halo rows a tile shares with its vertical neighbour have usually left the L2 by the time that neighbour runs; the XCD-chunked
// order that would keep them there lost to its HBM-channel concentration).  Here a block owns a 16-column strip of one (sample, 32-channel
// block) and WALKS DOWN it, 16 output rows per iteration, over a ring of 22 input rows in LDS: every input row of the strip leaves HBM once,
// whatever the caches do (only the 6 halo COLUMNS of a strip are read twice, by its neighbour strips: 22 / 16 from L2, 19 - 22 of 16 real).
// The 16 new rows of iteration t + 1 are requested BEFORE the arithmetic of iteration t (64 registers in flight per lane) and enter the ring
// after it; the outputs of iteration t leave after that refill, so the refill waits for its loads only, not for stores.  Same thread map,
// same per-output operation order as the tile kernel (bit-identical outputs); one statistics partial per strip instead of per tile.
// fp32 tensors with 32-channel blocks (NV = 8) and the split-precision tier's plane output only: that is where the bytes are.
constexpr int ST_W = 16, ST_R = 16, ST_HC = ST_W + 6, ST_HR = ST_R + 6, ST_NT = 256;
constexpr int ST_ROWB = ST_HC * 8 * 16;                 // bytes per ring row: 22 pixels x 128 B
constexpr int ST_XS = ST_HR * ST_ROWB;                  // 61 952
constexpr int ST_LDS = ST_XS + 49 * 32 * 4 + 64;
static_assert(LT_SR == 8, "the strip kernel's thread map is the tile kernel's at eight rows per thread");

__global__ __launch_bounds__(ST_NT, 2) void dwconv7_strip_kernel(const ds_dwconv_params p, int strips_w, int ncblk, int hparts, int rows_per_part, int order) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    char* const xs = dsm;                                                               // ring [22 rows][22 px][8 x 16 B]
    float* const wsm = reinterpret_cast<float*>(dsm + ST_XS);                           // [49][32]
    float* const red = wsm + 49 * 32;
    const int tid = threadIdx.x;
    const int nwg = gridDim.x;
    int wid = blockIdx.x;
    if ((order & 1) && (nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);       // XCD-chunked: consecutive items on one XCD
    int sw, cblk, hp, b;
    if (order & 2) {                       // strip fastest: the strips of one (sample, channel block) are neighbours in the order
        sw = wid % strips_w; cblk = (wid / strips_w) % ncblk; hp = (wid / (strips_w * ncblk)) % hparts; b = wid / (strips_w * ncblk * hparts);
    } else {                               // channel block fastest (the tile kernel's order)
        cblk = wid % ncblk; sw = (wid / ncblk) % strips_w; hp = (wid / (strips_w * ncblk)) % hparts; b = wid / (strips_w * ncblk * hparts);
    }
    const int bix = (hp * strips_w + sw) * ncblk + cblk;      // slot in the sample's statistics partials (order-independent sum)
    const int w0 = sw * ST_W, c0 = cblk * 32;
    const int hbeg = hp * rows_per_part, hend = min(p.H, hbeg + rows_per_part);
    const int ntile = (hend - hbeg + ST_R - 1) / ST_R;
    const int C = p.C0 + p.C1;
    const float* base;
    int Cs, cc, Hs, Ws, oh, ow;
    if (c0 < p.C0) {
        base = reinterpret_cast<const float*>(p.src0) + (size_t)b * p.H * p.W * p.C0;
        Cs = p.C0; cc = c0; Hs = p.H; Ws = p.W; oh = 0; ow = 0;
    } else {
        base = reinterpret_cast<const float*>(p.src1) + (size_t)b * p.H1 * p.W1 * p.C1;
        Cs = p.C1; cc = c0 - p.C0; Hs = p.H1; Ws = p.W1; oh = p.off_h1; ow = p.off_w1;
    }
    // ---- staging map: thread -> (halo column hc = tid >> 3 < 22, 16-byte piece v = tid & 7); row `it` of a batch of rows.  Waves 0, 1 are
    // fully active, wave 2 three quarters, wave 3 not at all (its branch is wave-uniform)
    const int hc = tid >> 3, sv = tid & 7;
    const bool stager = hc < ST_HC;
    const int wi = w0 + hc - 3 - ow;
    const bool col_ok = stager && (unsigned)wi < (unsigned)Ws;
    const unsigned colbyte = (unsigned)((wi * Cs + cc + sv * 4) * 4);       // byte offset inside an image row (garbage when !col_ok: masked below)
    const unsigned rowpitch = (unsigned)(Ws * Cs * 4);
#if !DS_BOUNDS
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)((size_t)Hs * Ws * Cs * 4), 0x00020000);
#endif
    auto load_rows = [&](u32x4* dst, int row0, auto nc) {                    // image rows row0 .. row0 + n - 1 (this source's coordinates: - oh)
        constexpr int n = decltype(nc)::value;
#pragma unroll
        for (int it = 0; it < n; ++it) {
            const int hi = row0 + it - oh;
            const unsigned bad = (unsigned)(!col_ok) | (unsigned)((unsigned)hi >= (unsigned)Hs);
#if DS_BOUNDS
            dst[it] = u32x4{0u, 0u, 0u, 0u};
            if (!bad) dst[it] = DS_LD(u32x4, reinterpret_cast<const char*>(base) + (size_t)hi * rowpitch + colbyte, c0 < p.C0 ? DS_BX_SRC0 : DS_BX_SRC1);
#else
            const unsigned off = (unsigned)hi * rowpitch + colbyte;
            dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((off & 0x7fffffffu) | (bad << 31)), 0, 0);
#endif
        }
    };
    auto store_rows = [&](const u32x4* src, int slot0, auto nc) {            // ring slots (slot0 + it) mod 22
        constexpr int n = decltype(nc)::value;
#pragma unroll
        for (int it = 0; it < n; ++it) {
            int sl = slot0 + it;
            sl = sl >= ST_HR ? sl - ST_HR : sl;
            *reinterpret_cast<u32x4*>(xs + sl * ST_ROWB + tid * 16) = src[it];
        }
    };
    auto lds_barrier = [&]() {          // LDS-only synchronisation: no wait for global loads / stores in flight (a __syncthreads() drains vmcnt)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using N6 = std::integral_constant<int, 6>;
    using N16 = std::integral_constant<int, ST_R>;

    // ---- prologue: weights, bias, the first 22 rows
    u32x4 hv[ST_R];
    u32x4 h6[6];
    if (stager) {
        load_rows(h6, hbeg - 3, N6{});
        load_rows(hv, hbeg + 3, N16{});
    }
    {
        constexpr int WP = 49 * 32 / 4, WIT = (WP + ST_NT - 1) / ST_NT;     // 392 pieces of 16 B: two per thread
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int wp = tid + k * ST_NT, wtap = wp >> 3, wj = wp & 7;
            if (wp < WP) *reinterpret_cast<f32x4*>(wsm + 4 * wp) = DS_LD(f32x4, p.wt + (size_t)wtap * C + c0 + 4 * wj, DS_BX_W);
        }
    }
    const int cv = tid & 7, wl = (tid >> 3) & 15, strip = tid >> 7;
    const int c = c0 + cv * 4;
    float init[4];
    {
        const f32x4 b4 = DS_LD(f32x4, p.bias + c, DS_BX_BIAS);
        f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
        if (p.tbias) t4 = DS_LD(f32x4, p.tbias + (size_t)b * p.tb_stride + c, DS_BX_AUX1);
#pragma unroll
        for (int j = 0; j < 4; ++j) init[j] = b4[j] + t4[j];
    }
    if (stager) {
        store_rows(h6, 0, N6{});
        store_rows(hv, 6, N16{});
    }
    lds_barrier();

    float s1 = 0.f, s2 = 0.f;
    int rbase = 0;                                               // ring slot of the tile's first input row (output row - 3)
    const int w = w0 + wl;
    const int colb = wl * 128 + cv * 16;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int htop = hbeg + t * ST_R;
        const bool more = t + 1 < ntile;
        if (more && stager && !(order & 8)) load_rows(hv, htop + ST_R + 3, N16{});          // rows 6 .. 21 of the next tile  (order bits 2..4: timing ablations, wrong results)
        // byte offsets of this thread's 14 input rows in the ring
        int roff[LT_SR + 6];
#pragma unroll
        for (int r = 0; r < LT_SR + 6; ++r) {
            int sl = rbase + strip * LT_SR + r;
            sl = sl >= ST_HR ? sl - ST_HR : sl;
            sl = sl >= ST_HR ? sl - ST_HR : sl;
            roff[r] = sl * ST_ROWB + colb;
        }
        float acc[LT_SR][4];
#pragma unroll
        for (int o = 0; o < LT_SR; ++o)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[o][v] = init[v];
        const int ndw = (order & 16) ? 1 : 7;
#pragma unroll 1
        for (int dw = 0; dw < ndw; ++dw) {   // not unrolled: keeps only one tap column of weights + inputs live
            float wv[7][4];
#pragma unroll
            for (int dh = 0; dh < 7; ++dh) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(wsm + (dh * 7 + dw) * 32 + cv * 4);
                wv[dh][0] = t4[0]; wv[dh][1] = t4[1]; wv[dh][2] = t4[2]; wv[dh][3] = t4[3];
            }
#pragma unroll
            for (int r = 0; r < LT_SR + 6; ++r) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(xs + roff[r] + dw * 128);
#pragma unroll
                for (int dh = 0; dh < 7; ++dh) {
                    const int o = r - dh;
                    if (o >= 0 && o < LT_SR) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[o][v] = fmaf(x[v], wv[dh][v], acc[o][v]);
                    }
                }
            }
        }
        lds_barrier();                                           // every wave has read what it needs of this tile's rows
        if (more) {
            int ws = rbase + ST_R + 6;                           // slot of the next tile's row 6 = this tile's row 22 -> (rbase + 22) mod 22 = rbase
            ws = ws >= ST_HR ? ws - ST_HR : ws;
            ws = ws >= ST_HR ? ws - ST_HR : ws;
            if (stager) store_rows(hv, ws, N16{});
            rbase = rbase + ST_R >= ST_HR ? rbase + ST_R - ST_HR : rbase + ST_R;
        }
        // outputs of this tile: hi / lo bf16 planes of a 2C-channel image (DS_CONV_F_SPLIT_IN), or fp32
#pragma unroll
        for (int o = 0; o < LT_SR; ++o) {
            const int h = htop + strip * LT_SR + o;
            if (h < hend && w < p.W && !((order & 4) && acc[o][0] != 12345.f)) {
                if (p.out_split) {
                    bf16* o2 = reinterpret_cast<bf16*>(p.out) + ((size_t)b * p.H * p.W + (size_t)(h * p.W + w)) * (2 * C) + c;
                    uint2 hi, lo;
                    ds_split2(acc[o][0], acc[o][1], hi.x, lo.x);
                    ds_split2(acc[o][2], acc[o][3], hi.y, lo.y);
                    // (r05 ablation: hi | lo of the block's 32 channels as ONE 128-byte line per pixel instead of two 64-byte runs in two planes:
                    // no consistent gain, 613 vs 595 and 591 vs 623 us on the two layer shapes tried — the plane layout stays)
                    DS_ST(bf16x4, o2, DS_BX_OUT, __builtin_bit_cast(bf16x4, hi));
                    DS_ST(bf16x4, o2 + C, DS_BX_OUT, __builtin_bit_cast(bf16x4, lo));
                } else {
                    float* o4 = reinterpret_cast<float*>(p.out) + ((size_t)b * p.H * p.W + (size_t)(h * p.W + w)) * C + c;
                    DS_ST(f32x4, o4, DS_BX_OUT, (f32x4{acc[o][0], acc[o][1], acc[o][2], acc[o][3]}));
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    s1 += acc[o][v];
                    s2 += acc[o][v] * acc[o][v];
                }
            }
        }
        if (more) lds_barrier();                                 // the refilled rows are visible
    }
    __syncthreads();
    if (p.stats_part) block_stats_write(s1, s2, red, p.stats_part + ((size_t)b * (hparts * strips_w * ncblk) + bix) * 2);
}

// ------------------------------------------------------------------------------------------------ dwconv7 on MFMA
// The VALU stencil above needs 49 fma + conversions per output and is issue-bound at ~2x its own floor.  Here the
// horizontal part of the stencil becomes a banded (Toeplitz) matrix, so one channel's 16x16 output block is
//     out[h][w] = sum_{dh} sum_{w'} x[h+dh][w'] * T_c[(dh,w')][w],   T_c[(dh,w')][w] = k_c[dh][w'-w]  (0 <= w'-w < 7)
// = A[16 x 168] . T_c[168 x 16] -> 6 x v_mfma_f32_16x16x32_bf16 (K padded to 192).  3.9x more MACs than the
// stencil, on a pipe that is 16x faster, with no bf16->fp32 conversions.
//   LDS holds the input halo PLANAR ([channel][row][cols] bf16) so that a lane's 8 consecutive k (= 8 adjacent columns of one
//   channel/row) are one 16-byte read; T_c fragments are precomputed per channel (ds_pack_dw_weight_mfma).
//   K order: MFMA ks, lane group kq -> (dh, wg) = (4 * (ks & 1) + kq, ks >> 1): the four lane groups of one read differ by whole plane
//   rows only, so the 16 lanes of every ds_read_b128 group touch distinct (or identical) rows and, with the 80- / 48-byte row pitch,
//   distinct bank slots (the order (dh, wg) = (G / 3, G % 3) made every read a 2-way conflict).  dh = 7 is padding: row 6 again, zero weights.
// (The first generation of this kernel — one block per tile and 32 channels, two blocks per CU, 2.65 TB/s — is in the history up to round 3.)
constexpr int MF_CB = 32;

// ------------------------------------------------------------------------------------------------ dwconv7 on MFMA, persistent blocks
// One block per tile (the first generation) is bound by latency and instruction issue, not by a
// pipe (profiles/r03_dw_pmc.txt: 48 % of wave cycles in s_waitcnt, MFMA pipe 11 % busy, LDS 42 %): every block pays one exposed memory round
// trip for its halo and re-fetches 24 KB of Toeplitz fragments per wave (192 KB per block against 53 KB of input).  Here ONE block per CU
// walks over CHUNKS of up to 8 tiles of one (sample, 32-channel block):
//   * the fragments of the wave's 2 channels stay in registers for the whole chunk (48 VGPRs; 16 waves per block, 128 registers per lane);
//   * the halo of tile t+1 is requested BEFORE the MFMA phase of tile t (16 VGPRs) and written into the other of two LDS plane sets after
//     tile t's output has left: the memory round trip hides behind a whole tile of work;
//   * a thread stages FOUR adjacent pixels of 8 channels (880 slots, one per thread): 8-byte LDS writes, half as many as the pixel-pair form,
//     and the plane sets are skewed by 64 bytes per 8 planes (whole planes apart the four lanes of a pixel quad hit one bank);
//   * the output tile is staged in its own 32 KB (16-byte chunks XOR-swizzled by the pixel column: the 8-byte writes of a wave were 8-way
//     bank conflicts) and leaves as whole 64-byte pixel rows;
//   * GroupNorm partials: one (sum, sum of squares) pair per chunk instead of one per tile.
// Chunks are dealt round-robin to the blocks in the XCD-chunked block order, channel block fastest: the blocks of one XCD work on the
// channel blocks of the same tiles at the same time (they share every 128-byte line of the input).
// two tile shapes of 512 pixels (two 16 x 16 MFMA blocks): WIDE 16 rows x 32 columns, TALL 32 rows x 16 columns for images at most 16 wide
template <bool TALL> struct M2 {
    static constexpr int W = TALL ? 16 : 32, H = TALL ? 32 : 16;
    static constexpr int HR = H + 6;                                   // halo rows
    static constexpr int HC = TALL ? 24 : 40;                          // plane row pitch in elements (22 / 38 used): whole pixel quads
    static constexpr int PLANE = HR * HC, SKEW = 32;
    static constexpr int BLK2 = TALL ? 16 * HC : 16;                   // element offset of the second 16 x 16 block inside a plane
    static constexpr int PBYTES = (MF_CB * PLANE + 3 * SKEW) * 2;      // one plane set: 56512 / 58560 bytes
    static constexpr int OFF_O = 2 * PBYTES, OBYTES = H * W * MF_CB * 2, OFF_RED = OFF_O + OBYTES, LDS = OFF_RED + 64;
    static constexpr int SLOTS = HR * (HC / 4) * 4;                    // 880 / 912 (pixel quad, 8-channel group) pairs: one per thread
    static_assert(LDS <= 160 * 1024, "one block per CU");
};

struct Dw2Geo { int tiles_w, tiles, ncblk, tpc, nchunk, total; };
typedef __amdgpu_buffer_rsrc_t dw_rsrc_t;
__device__ __forceinline__ u32x4 dw_buf_ld16(dw_rsrc_t rs, const char* base, unsigned voff, int bounds_buf) {
#if DS_BOUNDS
    if (voff >= 0x80000000u || !ds_bx_ok(base + voff, bounds_buf, 16)) return u32x4{0u, 0u, 0u, 0u};
#endif
    (void)base; (void)bounds_buf;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
}

template <int NW, bool TALL>
__global__ __launch_bounds__(NW * 64, 1) void dwconv7_mfma2_kernel(const ds_dwconv_params p, const Dw2Geo g) {
    using G = M2<TALL>;
    constexpr int M2_W = G::W, M2_H = G::H, M2_HC = G::HC, M2_PLANE = G::PLANE, M2_SKEW = G::SKEW, M2_PBYTES = G::PBYTES, M2_OFF_O = G::OFF_O,
                  M2_OFF_RED = G::OFF_RED, M2_SLOTS = G::SLOTS;
    constexpr int NT = NW * 64, CPW = MF_CB / NW, SIT = (M2_SLOTS + NT - 1) / NT, OIT = M2_H * M2_W * 4 / NT;      // channels per wave, halo slots and output pieces per thread
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    float* red = reinterpret_cast<float*>(dsm + M2_OFF_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NB = gridDim.x;
    int lid = blockIdx.x;
    if (NB % 8 == 0) lid = (blockIdx.x & 7) * (NB >> 3) + (blockIdx.x >> 3);
    if (lid >= g.total) return;
    const int nmy = (g.total - lid + NB - 1) / NB;            // this block's chunks: lid, lid + NB, ...
    const int nt = nmy * g.tpc;                               // ... = nt tiles, walked as one sequence
    const int C = p.C0 + p.C1;
    const bf16* wexp = reinterpret_cast<const bf16*>(p.wexp);

    // ---- a tile of the sequence: channel block, sample, origin, source (the skip concat is two sources with an offset).  A chunk is
    // g.tpc consecutive (sample, tile) items of ONE channel block — part of a sample's tiles, or several whole samples where an image has
    // only one or two tiles; the divisions run once per chunk, inside a chunk the position advances incrementally (all scalar: wave-uniform)
    struct Tile { const char* base; dw_rsrc_t rs; int c, j, tile, th, tw, b, c0, h0, w0, Cs, cc, Hs, Ws, oh, ow, sbuf; };
    auto set_sample = [&](Tile& t) {
        if (t.c0 < p.C0) {
            t.base = reinterpret_cast<const char*>(p.src0) + (size_t)t.b * p.H * p.W * p.C0 * 2;
            t.Cs = p.C0; t.cc = t.c0; t.Hs = p.H; t.Ws = p.W; t.oh = 0; t.ow = 0; t.sbuf = DS_BX_SRC0;
        } else {
            t.base = reinterpret_cast<const char*>(p.src1) + (size_t)t.b * p.H1 * p.W1 * p.C1 * 2;
            t.Cs = p.C1; t.cc = t.c0 - p.C0; t.Hs = p.H1; t.Ws = p.W1; t.oh = p.off_h1; t.ow = p.off_w1; t.sbuf = DS_BX_SRC1;
        }
        // one sample of the source as a range-checked buffer: halo pixels outside the image carry an out-of-range offset and read as zeros
        // (no select on the loaded data: a select would make the wave wait for the halo right where it is requested)
        t.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(t.base), (short)0, t.Hs * t.Ws * t.Cs * 2, 0x00020000);
    };
    auto start_chunk = [&](int c) {
        Tile t;
        t.c = c;
        t.j = 0;
        const int cblk = c % g.ncblk, item = (c / g.ncblk) * g.tpc;
        t.b = item / g.tiles;
        t.tile = item - t.b * g.tiles;
        t.th = t.tile / g.tiles_w;
        t.tw = t.tile - t.th * g.tiles_w;
        t.h0 = t.th * M2_H;
        t.w0 = t.tw * M2_W;
        t.c0 = cblk * MF_CB;
        set_sample(t);
        return t;
    };
    auto next_tile = [&](Tile t) {
        if (t.j + 1 == g.tpc) return start_chunk(t.c + NB);
        ++t.j;
        if (++t.tile == g.tiles) {                             // next sample, same channels
            t.tile = 0; t.th = 0; t.tw = 0;
            ++t.b;
            set_sample(t);
        } else if (++t.tw == g.tiles_w) { t.tw = 0; ++t.th; }
        t.h0 = t.th * M2_H;
        t.w0 = t.tw * M2_W;
        return t;
    };

    // ---- halo staging slots of this thread (tile-independent): slot -> (8-channel group v, halo row hr, first column hc of a pixel quad)
    int s_hr[SIT], s_hc[SIT], s_lds[SIT];
    bool s_ok[SIT];
    const int sv = tid & 3;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int slot = tid + it * NT, qd = slot >> 2;
        s_ok[it] = slot < M2_SLOTS;
        s_hr[it] = qd / (M2_HC / 4);
        s_hc[it] = (qd - s_hr[it] * (M2_HC / 4)) * 4;
        s_lds[it] = ((sv * 8) * M2_PLANE + sv * M2_SKEW + s_hr[it] * M2_HC + s_hc[it]) * 2;      // bytes inside a plane set
    }
    u32x4 fv[SIT][4];
    auto issue_halo = [&](const Tile& t) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int hi = t.h0 + s_hr[it] - 3 - t.oh, wi = t.w0 + s_hc[it] - 3 - t.ow;
            // (arithmetic, not a select: any offset with bit 31 set is beyond the buffer.  The offset of a row above the image is garbage: every
            // offset is cut to 28 bits — a sample is far below 256 MB — so that bit 31 plus it plus 16 bytes cannot wrap around 2^32 into the buffer)
            const unsigned badr = (unsigned)(!s_ok[it]) | (unsigned)((unsigned)hi >= (unsigned)t.Hs);
            const unsigned o = (unsigned)((hi * t.Ws + wi) * t.Cs + t.cc + sv * 8) * 2u;       // (may be "negative": pixel quads straddle the left edge)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned bad = badr | (unsigned)((unsigned)(wi + e) >= (unsigned)t.Ws);
                fv[it][e] = dw_buf_ld16(t.rs, t.base, ((o + (unsigned)(e * t.Cs) * 2u) & 0x0fffffffu) | (bad << 31), t.sbuf);
            }
        }
    };
    auto fill_planes = [&](char* pl) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            if (s_ok[it]) {
                char* dst = pl + s_lds[it];
#pragma unroll
                for (int j = 0; j < 4; ++j) {                  // dword j of a pixel = channels 2j, 2j+1; a plane row gets 4 pixels = 8 bytes
                    const unsigned a = fv[it][0][j], b2 = fv[it][1][j], c2 = fv[it][2][j], d = fv[it][3][j];
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2_t*>(dst + (2 * j) * (M2_PLANE * 2)) =
                        u32x2_t{__builtin_amdgcn_perm(b2, a, 0x05040100u), __builtin_amdgcn_perm(d, c2, 0x05040100u)};
                    *reinterpret_cast<u32x2_t*>(dst + (2 * j + 1) * (M2_PLANE * 2)) =
                        u32x2_t{__builtin_amdgcn_perm(b2, a, 0x07060302u), __builtin_amdgcn_perm(d, c2, 0x07060302u)};
                }
            }
        }
    };

    const int m = lane & 15, kq = lane >> 4;
    int aoff[6];                                               // A fragment offsets (elements): the K order above
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int dh = 4 * (ks & 1) + kq, wg = ks >> 1;
        aoff[ks] = (m + (dh > 6 ? 6 : dh)) * M2_HC + 8 * wg;
    }
    bf16x8 wv[CPW][6];
    float addv[CPW];
    auto load_addv = [&](const Tile& t) {                      // bias (+ this sample's time bias) of this wave's channels
        const int cb = t.c0 + wave * CPW;
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            addv[k] = DS_LD(float, p.bias + cb + k, DS_BX_BIAS);
            if (p.tbias) addv[k] += DS_LD(float, p.tbias + (size_t)t.b * p.tb_stride + cb + k, DS_BX_AUX1);
        }
    };
    auto load_chunk_consts = [&](const Tile& t) {              // Toeplitz fragments of this wave's channels
        const int cb = t.c0 + wave * CPW;
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            const bf16* we = wexp + ((size_t)(cb + ci) * 6 * 64 + lane) * 8;
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) wv[ci][ks] = DS_LD(bf16x8, we + ks * 64 * 8, DS_BX_AUX0);
        }
        load_addv(t);
    };

    Tile cur = start_chunk(lid);
    issue_halo(cur);
    load_chunk_consts(cur);
    fill_planes(dsm);
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    char* const ot = dsm + M2_OFF_O;
    // (the body is instantiated twice — with and without a next tile — so that inside the loop nothing is conditional on it: with an
    // `if (more)` around the halo request and another around the fill, the compiler's wait-count bookkeeping merges the two paths, takes the
    // halo loads for possibly still pending at the back edge and guards the next request with waits that in fact wait for the output stores)
    auto tile_body = [&](const int u, auto more_t) {
        constexpr bool more = decltype(more_t)::value;
        char* const pl = dsm + (u & 1) * M2_PBYTES;
        Tile nxt = cur;
        if constexpr (more) {
            nxt = next_tile(cur);
            issue_halo(nxt);                                   // in flight during this tile's MFMA phase
        }
        // ---- CPW channels x two 16 x 16 blocks: 12 MFMAs per channel
        float outv[2][4][CPW];                                 // [column block][row][channel]
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            const int cl = wave * CPW + ci;
            const bf16* plane = reinterpret_cast<const bf16*>(pl) + cl * M2_PLANE + (cl >> 3) * M2_SKEW;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(plane + aoff[ks]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(plane + aoff[ks] + G::BLK2);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wv[ci][ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wv[ci][ks], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                outv[0][r][ci] = acc0[r] + addv[ci];
                outv[1][r][ci] = acc1[r] + addv[ci];
            }
        }
        // C/D layout of 16x16x32: col = lane & 15 (w), row = (lane >> 4) * 4 + r (h).  Statistics from the fp32 values.
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (cur.h0 + (TALL ? 16 * wb : 0) + kq * 4 + r < p.H && cur.w0 + (TALL ? 0 : 16 * wb) + m < p.W) {
#pragma unroll
                    for (int v = 0; v < CPW; ++v) {
                        s1 += outv[wb][r][v];
                        s2 = fmaf(outv[wb][r][v], outv[wb][r][v], s2);
                    }
                }
        // ---- output tile -> LDS [rows][cols][32 ch] bf16 (64 B per pixel; 16-byte chunk q of column col at q ^ ((col >> 1) & 3))
#pragma unroll
        for (int wb = 0; wb < 2; ++wb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                typedef __bf16 bf16xc_t __attribute__((ext_vector_type(CPW)));
                bf16xc_t pk;
#pragma unroll
                for (int v = 0; v < CPW; ++v) pk[v] = (bf16)outv[wb][r][v];
                const int col = (TALL ? 0 : 16 * wb) + m, row = (TALL ? 16 * wb : 0) + kq * 4 + r, cbyte = wave * CPW * 2;          // this wave's channels: bytes cbyte .. of the pixel's 64
                *reinterpret_cast<bf16xc_t*>(ot + (row * M2_W + col) * 64 + (((cbyte >> 4) ^ ((col >> 1) & 3)) * 16) + (cbyte & 15)) = pk;
            }
        __syncthreads();                                       // the output tile is complete; every wave is done with this plane set
        const bool chunk_end = cur.j + 1 == g.tpc, sample_end = chunk_end || cur.tile + 1 == g.tiles;
        if (sample_end && p.stats_part) {                      // (block-uniform; one barrier inside): one partial per (sample, chunk, channel block)
            const int nps = g.tiles >= g.tpc ? g.tiles / g.tpc : 1, kk = g.tiles >= g.tpc ? cur.tile / g.tpc : 0;
            block_stats_write(s1, s2, red, p.stats_part + ((size_t)cur.b * (nps * g.ncblk) + kk * g.ncblk + cur.c0 / MF_CB) * 2);
            s1 = 0.f;
            s2 = 0.f;
        }
        // the next tile's halo has had the whole MFMA phase to arrive: into the other plane set first, THEN this tile's output — its
        // stores stay in flight across the barrier and the next tile (only the LDS reads that feed them must be done before it)
        if constexpr (more) {
            if (chunk_end || sample_end) {
                if (chunk_end) load_chunk_consts(nxt);
                else load_addv(nxt);
                // consumed (= waited for) inside the branch, once per chunk: loads and stores retire out of order with respect to each other,
                // so with these left pending at the back edge the first MFMA of EVERY tile would be guarded by s_waitcnt vmcnt(0) — which
                // in the common path waits for the previous tile's output stores
#pragma unroll
                for (int ci = 0; ci < CPW; ++ci)
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(wv[ci][ks]));
#pragma unroll
                for (int ci = 0; ci < CPW; ++ci) asm volatile("" : "+v"(addv[ci]));
            }
            fill_planes(dsm + ((u + 1) & 1) * M2_PBYTES);
        }
        {
            bf16* outp = reinterpret_cast<bf16*>(p.out) + (size_t)cur.b * p.H * p.W * C;
#pragma unroll
            for (int k = 0; k < OIT; ++k) {
                const int piece = tid + k * NT, px = piece >> 2, q = piece & 3;   // 4 consecutive lanes = one pixel's 64 bytes
                const int col = px % M2_W, h = cur.h0 + px / M2_W, w = cur.w0 + col;
                if (h < p.H && w < p.W) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ot + px * 64 + ((q ^ ((col >> 1) & 3)) * 16));
                    DS_ST(u32x4, outp + ((size_t)(h * p.W + w) * C + cur.c0 + q * 8), DS_BX_OUT, v);
                }
            }
        }
        __syncthreads();                                       // next plane set complete, output staging free
        cur = nxt;
    };
    for (int u = 0; u + 1 < nt; ++u) tile_body(u, std::true_type{});
    tile_body(nt - 1, std::false_type{});
}

static bool dw_tall(const ds_dwconv_params* p) { return p->W <= 16; }
static Dw2Geo dw2_geo(const ds_dwconv_params* p) {
    const int tw = dw_tall(p) ? 16 : 32, th = dw_tall(p) ? 32 : 16;
    Dw2Geo g;
    g.tiles_w = (p->W + tw - 1) / tw;
    g.tiles = g.tiles_w * ((p->H + th - 1) / th);
    g.ncblk = (p->C0 + p->C1) / MF_CB;
    // chunk = tpc consecutive (sample, tile) items of one channel block: a divisor of a sample's tiles, or whole samples (tpc a multiple
    // of the tiles of one) — never a run that ends inside one sample and starts inside the next
    // The longest such chunk (<= 8 items) that still leaves two chunks per CU: small batches get short chunks (parallelism before amortisation)
    g.tpc = 1;
    for (int d = 2; d <= 8; ++d)
        if ((g.tiles % d == 0 || (d % g.tiles == 0 && p->B % (d / g.tiles) == 0)) && (long long)p->B * g.tiles / d * g.ncblk >= 512) g.tpc = d;
    g.nchunk = p->B * g.tiles / g.tpc;                        // chunks per channel block
    g.total = g.nchunk * g.ncblk;
    return g;
}
static int dw2_parts(const Dw2Geo& g) { return (g.tiles >= g.tpc ? g.tiles / g.tpc : 1) * g.ncblk; }   // GroupNorm partials per sample

__global__ void pack_dw_mfma_kernel(const float* w, int C, bf16* dst) {
    // dst[c][ks][lane][j]: B operand of 16x16x32 — lane = (n = lane & 15, kq = lane >> 4), k = ks*32 + kq*8 + j
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * 6 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 6, c = i / (6 * 64 * 8);
    const int n = lane & 15, kq = lane >> 4;
    const int dh = 4 * (ks & 1) + kq, wg = ks >> 1, wp = 8 * wg + j, dw = wp - n;      // K order of dwconv7_mfma2_kernel's aoff[]
    float v = 0.f;
    if (dh < 7 && dw >= 0 && dw < 7) v = w[(size_t)c * 49 + dh * 7 + dw];
    dst[i] = (bf16)v;
}

__global__ void pack_dw_kernel(const float* w, int C, float* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 49*C, dst[tap][c] = w[c][tap]
    if (i < 49 * C) dst[i] = w[(size_t)(i % C) * 49 + i / C];
}

// ------------------------------------------------------------------------------------------------ GroupNorm
__global__ void gn_finalize_kernel(const float* part, int parts, double count, float eps, float* ab) {
    __shared__ double r1[256], r2[256];
    const int b = blockIdx.x;
    double a = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < parts; i += blockDim.x) {
        a += (double)part[((size_t)b * parts + i) * 2];
        q += (double)part[((size_t)b * parts + i) * 2 + 1];
    }
    r1[threadIdx.x] = a;
    r2[threadIdx.x] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            r1[threadIdx.x] += r1[threadIdx.x + s];
            r2[threadIdx.x] += r2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = r1[0] / count;
        double var = r2[0] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        ab[2 * b] = (float)rstd;
        ab[2 * b + 1] = (float)(rstd * mean);
    }
}

// direct statistics for (sample, group): one block per (b, g); double accumulation of fp32 partials
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* x, int HW, int C, int G, float eps, float* ab) {
    __shared__ double r1[256], r2[256];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G;
    const T* xb = x + (size_t)b * HW * C + g * cg;
    double a = 0.0, q = 0.0;
    const long n = (long)HW * cg;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const long pix = i / cg;
        const int c = i - pix * cg;
        const float v = to_f32(xb[pix * C + c]);
        a += v;
        q += (double)v * v;
    }
    r1[threadIdx.x] = a;
    r2[threadIdx.x] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            r1[threadIdx.x] += r1[threadIdx.x + s];
            r2[threadIdx.x] += r2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = r1[0] / (double)n;
        double var = r2[0] / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        ab[2 * blockIdx.x] = (float)rstd;
        ab[2 * blockIdx.x + 1] = (float)(rstd * mean);
    }
}

// Streaming statistics for GroupNorm(G, C) over channels-last tensors: the per-(sample, group) kernel above reads a group's
// C/G channels of every pixel as scalars (10-byte pieces of a 160-byte pixel for the VQGAN decoder's last stage: 2.3 TB/s on
// a 1.3 GB tensor).  Here every block streams a contiguous range of one sample's pixels with 16-byte vector loads, each
// thread owning a FIXED channel vector (blockDim is a multiple of the C/V vectors of a pixel) so its partial sums stay in
// registers; per-channel (sum, sumsq) block partials go to a workspace and gn_stats_finish folds channels into groups in
// float64.  No atomics: bit-reproducible.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_stream_kernel(const T* x, int HW, int C, int pix_per_blk, float* part) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    float* sm = reinterpret_cast<float*>(gsm);                 // [rows][CV][V][2]
    const int CV = C / V, rows = blockDim.x / CV, b = blockIdx.y;
    const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
    const int p0 = blockIdx.x * pix_per_blk, p1 = min(HW, p0 + pix_per_blk);
    float s1[V], s2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s1[v] = s2[v] = 0.f;
    const T* xb = x + (size_t)b * HW * C + cv * V;
    // four pixels of the thread in flight (one load per iteration left a 240-thread block with 240 x 16 bytes in flight: 1.9 TB/s)
    int pix = p0 + row;
    for (; pix + 3 * rows < p1; pix += 4 * rows) {
        float f[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) Vec16<T>::load(xb + (size_t)(pix + k * rows) * C, f[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                s1[v] += f[k][v];
                s2[v] = fmaf(f[k][v], f[k][v], s2[v]);
            }
    }
    for (; pix < p1; pix += rows) {
        float f[V];
        Vec16<T>::load(xb + (size_t)pix * C, f);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            s1[v] += f[v];
            s2[v] = fmaf(f[v], f[v], s2[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        sm[((row * CV + cv) * V + v) * 2] = s1[v];
        sm[((row * CV + cv) * V + v) * 2 + 1] = s2[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 2; i += blockDim.x) {   // i = (channel, which)
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += sm[r * C * 2 + i];
        part[(((size_t)b * gridDim.x + blockIdx.x) * C) * 2 + i] = a;
    }
}
__global__ void gn_stats_finish_kernel(const float* part, int nblk, int C, int G, double count, float eps, float* ab) {
    // one wave per (sample, group)
    const int b = blockIdx.x / G, g = blockIdx.x % G, cg = C / G, lane = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int i = lane; i < nblk * cg; i += 64) {
        const int blk = i / cg, c = g * cg + i % cg;
        const float* pp = part + (((size_t)b * nblk + blk) * C + c) * 2;
        a += (double)pp[0];
        q += (double)pp[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) {
        const double mean = a / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        ab[2 * blockIdx.x] = (float)rstd;
        ab[2 * blockIdx.x + 1] = (float)(rstd * mean);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const ds_gn_apply_params p, size_t nvec) {
    constexpr int V = Vec16<T>::N;
    const int CV = p.C / V;
    const int cg = p.C / p.G;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = i % CV;
        const size_t pix = i / CV;
        const int b = pix / p.HW;
        const int c = cv * V;
        float x[V], o[V], gm[V], bt[V], r[V];
        vec16_load<T>(reinterpret_cast<const T*>(p.x) + i * V, x, DS_BX_SRC0);
        if (p.res) vec16_load<T>(reinterpret_cast<const T*>(p.res) + i * V, r, DS_BX_RES);
#pragma unroll
        for (int v = 0; v < V; v += 4) {   // per-channel affine as 16-byte loads (C is a multiple of V, so 16-B aligned)
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + c + v);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + c + v);
            gm[v] = g4[0]; gm[v + 1] = g4[1]; gm[v + 2] = g4[2]; gm[v + 3] = g4[3];
            bt[v] = b4[0]; bt[v + 1] = b4[1]; bt[v + 2] = b4[2]; bt[v + 3] = b4[3];
        }
        const float* abp = p.gn_ab + (size_t)b * p.G * 2;
        float a0 = abp[0], am0 = abp[1];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float a = a0, am = am0;
            if (p.G > 1) {
                const int g = (c + v) / cg;
                a = abp[2 * g];
                am = abp[2 * g + 1];
            }
            float y = (x[v] * a - am) * gm[v] + bt[v];
            y = act_apply(y, p.act);
            if (p.cbias) y += p.cbias[(size_t)b * p.cb_stride + c + v];
            if (p.res) y += r[v];
            o[v] = y;
        }
        vec16_store<T>(reinterpret_cast<T*>(p.out) + i * V, o, DS_BX_OUT);
    }
}

// GroupNorm(G, C) apply for G > 1 (the VQGAN's Normalize, VQGAN.py:12-27) with the per-channel (scale, shift) of the block's sample tabulated
// once in LDS and each thread on a FIXED channel vector (blockDim = CV x rows): y = act(x * scale + shift) (+ cbias + res).  The generic kernel
// above divides (c + v) / (C / G) and fetches the group's (rstd, rstd * mean) per ELEMENT behind a runtime activation switch with libm's expf:
// 155 us for the 128 x 64 x 160 tensor of the decoder at batch 64 (2.2 TB/s).
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_table_kernel(const ds_gn_apply_params p, int pix_per_blk) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float gtab[];       // [C][2]
    const int CV = p.C / V, rows = blockDim.x / CV, b = blockIdx.y, cg = p.C / p.G;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const int g = c / cg;
        const float a = p.gn_ab[((size_t)b * p.G + g) * 2], am = p.gn_ab[((size_t)b * p.G + g) * 2 + 1], gm = p.gamma[c];
        gtab[2 * c] = a * gm;
        gtab[2 * c + 1] = p.beta[c] - am * gm + (p.cbias ? p.cbias[(size_t)b * p.cb_stride + c] : 0.f) * (ACT == DS_ACT_NONE ? 1.f : 0.f);
    }
    __syncthreads();
    const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
    float sc[V], sh[V], cb[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        sc[v] = gtab[2 * (cv * V + v)];
        sh[v] = gtab[2 * (cv * V + v) + 1];
        cb[v] = (p.cbias && ACT != DS_ACT_NONE) ? p.cbias[(size_t)b * p.cb_stride + cv * V + v] : 0.f;   // (added AFTER the activation, as above)
    }
    const int p0 = blockIdx.x * pix_per_blk, p1 = min(p.HW, p0 + pix_per_blk);
    const size_t base = (size_t)b * p.HW * p.C + cv * V;
    const T* xb = reinterpret_cast<const T*>(p.x) + base;
    const T* rb = p.res ? reinterpret_cast<const T*>(p.res) + base : nullptr;
    T* ob = reinterpret_cast<T*>(p.out) + base;
    const bool has_res = rb != nullptr;
    auto one = [&](const float (&x)[V], const float (&r)[V], int pix) {
        float o[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float y = fmaf(x[v], sc[v], sh[v]);
            if (ACT == DS_ACT_SILU) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * y));
            else if (ACT == DS_ACT_RELU) y = fmaxf(y, 0.f);
            else if (ACT == DS_ACT_GELU) y = gelu_fast(y);
            y += cb[v];
            if (has_res) y += r[v];
            o[v] = y;
        }
        vec16_store<T>(ob + (size_t)pix * p.C, o, DS_BX_OUT);
    };
    int pix = p0 + row;
    for (; pix + rows < p1; pix += 2 * rows) {                         // two pixels of the thread in flight
        float x0[V], x1[V], r0[V] = {}, r1[V] = {};
        vec16_load<T>(xb + (size_t)pix * p.C, x0, DS_BX_SRC0);
        vec16_load<T>(xb + (size_t)(pix + rows) * p.C, x1, DS_BX_SRC0);
        if (rb) {
            vec16_load<T>(rb + (size_t)pix * p.C, r0, DS_BX_RES);
            vec16_load<T>(rb + (size_t)(pix + rows) * p.C, r1, DS_BX_RES);
        }
        one(x0, r0, pix);
        one(x1, r1, pix + rows);
    }
    if (pix < p1) {
        float x0[V], r0[V] = {};
        vec16_load<T>(xb + (size_t)pix * p.C, x0, DS_BX_SRC0);
        if (rb) vec16_load<T>(rb + (size_t)pix * p.C, r0, DS_BX_RES);
        one(x0, r0, pix);
    }
}

// one sample per blockIdx.y: the block reduces the producer's partials itself (GroupNorm(1, C) only)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_lazy_kernel(const ds_gn_apply_params p) {
    constexpr int V = Vec16<T>::N;
    const int CV = p.C / V, b = blockIdx.y;
    float a, am;
    gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, a, am);
    const size_t n = (size_t)p.HW * CV, base = (size_t)b * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        float x[V], o[V], r[V];
        vec16_load<T>(reinterpret_cast<const T*>(p.x) + (base + i) * V, x, DS_BX_SRC0);
        if (p.res) vec16_load<T>(reinterpret_cast<const T*>(p.res) + (base + i) * V, r, DS_BX_RES);
#pragma unroll
        for (int v = 0; v < V; v += 4) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + c + v);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + c + v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = (x[v + q] * a - am) * g4[q] + b4[q];
                y = act_apply(y, p.act);
                if (p.cbias) y += p.cbias[(size_t)b * p.cb_stride + c + v + q];
                if (p.res) y += r[v + q];
                o[v + q] = y;
            }
        }
        vec16_store<T>(reinterpret_cast<T*>(p.out) + (base + i) * V, o, DS_BX_OUT);
    }
}

// The U-Net's case of the kernel above (bf16, no activation, no channel bias; 96 / 192 / 384 / 768 channels): blocks of 192 threads
// = a whole number of pixels, so a thread keeps ONE channel vector — scale a * gamma and shift beta - a * mean * gamma live in
// registers, no per-element index arithmetic (the generic loop pays a 64-bit modulo and four table loads per 16 bytes) — and
// streams a contiguous pixel range four vectors at a time.  4.3 -> TB/s on 0.4 - 1.2 GB tensors, 16 launches per forward.
template <bool HAS_RES>
__global__ __launch_bounds__(192) void gn_apply_lazy_fast_kernel(const ds_gn_apply_params p, int pix_per_blk) {
    const int CV = p.C >> 3, rows = 192 / CV, b = blockIdx.y;
    const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
    float a, am;
    gn_from_partials(p.gn_part, p.gn_parts, p.gn_count, p.gn_eps, b, a, am);
    float sc[8], sh[8];
#pragma unroll
    for (int v = 0; v < 8; v += 4) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + cv * 8 + v);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + cv * 8 + v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[v + q] = a * g4[q];
            sh[v + q] = fmaf(-am, g4[q], b4[q]);
        }
    }
    const int p0 = blockIdx.x * pix_per_blk, p1 = min(p.HW, p0 + pix_per_blk);
    const size_t sbase = (size_t)b * p.HW * p.C + cv * 8;
    const bf16* xb = reinterpret_cast<const bf16*>(p.x) + sbase;
    const bf16* rb = reinterpret_cast<const bf16*>(p.res) + sbase;
    bf16* ob = reinterpret_cast<bf16*>(p.out) + sbase;
    constexpr int U = 4;
    for (int pix = p0 + row; pix < p1; pix += rows * U) {
        u32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = pix + u * rows;
            const int qc = q < p1 ? q : pix;                   // unconditional loads (clamped), predicated stores
            xv[u] = DS_LD(u32x4, xb + (size_t)qc * p.C, DS_BX_SRC0);
            if constexpr (HAS_RES) rv[u] = DS_LD(u32x4, rb + (size_t)qc * p.C, DS_BX_RES);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = pix + u * rows;
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = __uint_as_float(xv[u][e] << 16), hi = __uint_as_float(xv[u][e] & 0xffff0000u);
                lo = fmaf(lo, sc[2 * e], sh[2 * e]);
                hi = fmaf(hi, sc[2 * e + 1], sh[2 * e + 1]);
                if constexpr (HAS_RES) {
                    lo += __uint_as_float(rv[u][e] << 16);
                    hi += __uint_as_float(rv[u][e] & 0xffff0000u);
                }
                o[2 * e] = lo;
                o[2 * e + 1] = hi;
            }
            if (q < p1) vec16_store<bf16>(ob + (size_t)q * p.C, o, DS_BX_OUT);
        }
    }
}

}  // namespace

static bool dw_use_mfma(const ds_dwconv_params* p) {
    static const bool off = getenv("DS_DW_NO_MFMA") != nullptr;   // A/B switch: the LDS-tile stencil kernel
    const long long s0 = (long long)p->H * p->W * p->C0 * 2, s1 = (long long)p->H1 * p->W1 * p->C1 * 2;      // (28-bit halo offsets inside a sample)
    return p->dtype == DS_BF16 && p->wexp != nullptr && p->C0 % MF_CB == 0 && p->C1 % MF_CB == 0 && !off && s0 < (1ll << 28) && s1 < (1ll << 28);
}

static int lt_nv(const ds_dwconv_params* p) { return (p->dtype == DS_F32 && p->C0 % 32 == 0 && p->C1 % 32 == 0) ? 8 : 4; }
static bool dw_use_lds(const ds_dwconv_params* p) {
    const int CB = lt_nv(p) * (p->dtype == DS_BF16 ? 8 : 4);
    const long long es = p->dtype == DS_BF16 ? 2 : 4;
    // (a sample of either source below 2 GB: the halo loads carry "outside the image" in bit 31 of a 32-bit byte offset)
    const bool small = (long long)p->H * p->W * p->C0 * es < (1ll << 31) && (long long)p->H1 * p->W1 * p->C1 * es < (1ll << 31);
    return p->C0 % CB == 0 && p->C1 % CB == 0 && small;
}

// r05: the strip kernel (LDS ring walking down a 16-column strip) for the split-precision tier's depthwise layers.  Chosen by shape AND batch
// (it needs >= DS_DW_STRIP_MIN_ITEMS blocks, default 512 = two per CU; images of 256 rows may be cut into 2 - 4 row ranges to get there):
// its outputs are the tile kernel's bit for bit, its statistics partials are grouped differently — like the split-K choices of this tier
// (DESIGN §3), never taken in the fp32 parity tier unless DS_DW_STRIP=2 asks for it (tests).  DS_DW_STRIP=0: off.
struct DwStrip { int on, strips_w, ncblk, hparts, rows_per_part; };
static DwStrip dw_strip(const ds_dwconv_params* p) {
    static const int mode = getenv("DS_DW_STRIP") ? atoi(getenv("DS_DW_STRIP")) : 1;
    static const int min_items = getenv("DS_DW_STRIP_MIN_ITEMS") ? atoi(getenv("DS_DW_STRIP_MIN_ITEMS")) : 512;
    DwStrip g{0, 0, 0, 1, 0};
    const bool forced = p->strip == 1;
    if (p->strip == 2 || (mode == 0 && !forced) || p->dtype != DS_F32 || lt_nv(p) != 8 || !dw_use_lds(p) || (!p->out_split && mode != 2 && !forced) ||
        p->W < 16 || p->H < 2 * ST_R)
        return g;
    g.strips_w = (p->W + ST_W - 1) / ST_W;
    g.ncblk = (p->C0 + p->C1) / 32;
    const long n0 = (long)p->B * g.ncblk * g.strips_w;
    while (n0 * g.hparts < 2 * min_items && p->H / (2 * g.hparts) >= 2 * ST_R) g.hparts *= 2;
    if (n0 * g.hparts < min_items && !forced) return g;
    g.rows_per_part = ((p->H + g.hparts - 1) / g.hparts + ST_R - 1) / ST_R * ST_R;
    g.hparts = (p->H + g.rows_per_part - 1) / g.rows_per_part;
    g.on = 1;
    return g;
}

extern "C" int ds_dwconv_stats_parts(const ds_dwconv_params* p) {
    const int V = p->dtype == DS_BF16 ? 8 : 4;
    const int C = p->C0 + p->C1;
    if (dw_use_mfma(p)) {
        return dw2_parts(dw2_geo(p));
    }
    if (const DwStrip g = dw_strip(p); g.on) return g.hparts * g.strips_w * g.ncblk;
    if (dw_use_lds(p)) {
        const int nv = lt_nv(p), tw = 1 << lt_twl(p->W, nv), th = 2048 / nv / tw;
        return ((p->H + th - 1) / th) * ((p->W + tw - 1) / tw) * (C / (nv * V));
    }
    const long total = (long)((p->H + DW_TH - 1) / DW_TH) * p->W * (C / V);
    return (int)((total + DW_BLOCK - 1) / DW_BLOCK);
}

extern "C" int ds_dwconv7(const ds_dwconv_params* p, void* stream) {
    DS_REQUIRE(p && p->src0 && p->wt && p->bias && p->out, "dwconv7: null pointer");
    DS_REQUIRE(p->dtype == DS_F32 || p->dtype == DS_BF16, "dwconv7: dtype %d", p->dtype);
    const int V = p->dtype == DS_BF16 ? 8 : 4;
    const int C = p->C0 + p->C1;
    DS_REQUIRE(p->C0 > 0 && p->C0 % V == 0 && p->C1 % V == 0, "dwconv7: channels (%d,%d) must be multiples of %d", p->C0, p->C1, V);
    DS_REQUIRE(p->C1 == 0 || (p->src1 && p->H1 > 0 && p->W1 > 0), "dwconv7: second source incomplete");
    DS_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0, "dwconv7: empty problem");
    DS_REQUIRE(p->strip >= 0 && p->strip <= 2, "dwconv7: strip must be 0 (library's choice), 1 (strip kernel) or 2 (tile kernel), got %d", p->strip);
    DS_REQUIRE(!p->out_split || (p->dtype == DS_F32 && dw_use_lds(p)), "dwconv7: out_split needs the fp32 LDS-tile kernel (channels multiples of %d, samples below 2 GB)", 16);
    if (!ds_aligned16(p->src0) || !ds_aligned16(p->out) || !ds_aligned16(p->wt) || (p->C1 && !ds_aligned16(p->src1)))
        DS_FAIL(DS_EALIGN, "dwconv7: pointers must be 16-byte aligned");
    const int nstrip = (p->H + DW_TH - 1) / DW_TH;
    const int CV = C / V;
    const int blocks = ds_dwconv_stats_parts(p);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        const long long es = V == 8 ? 2 : 4;
        DsBxHost h(dw_use_mfma(p) ? DS_K_DWCONV_MFMA : (dw_use_lds(p) ? DS_K_DWCONV_LDS : DS_K_DWCONV));
        h.set(DS_BX_SRC0, p->src0, (long long)p->B * p->H * p->W * p->C0 * es);
        h.set(DS_BX_SRC1, p->C1 ? p->src1 : nullptr, (long long)p->B * p->H1 * p->W1 * p->C1 * es);
        h.set(DS_BX_W, p->wt, (long long)49 * C * 4);
        h.set(DS_BX_AUX0, p->wexp, (long long)C * 6 * 64 * 8 * 2);
        h.set(DS_BX_BIAS, p->bias, (long long)C * 4);
        h.set(DS_BX_AUX1, p->tbias, p->tbias ? ((long long)(p->B - 1) * p->tb_stride + C) * 4 : 0);
        h.set(DS_BX_OUT, p->out, (long long)p->B * p->H * p->W * C * es);
        h.set(DS_BX_STATS, p->stats_part, (long long)p->B * blocks * 2 * 4);
        h.publish(st);
    }
#endif
    if (dw_use_mfma(p)) {
        const Dw2Geo g = dw2_geo(p);
        const int nb = g.total < 256 ? g.total : 256;           // one block per CU
        // 16 waves of 2 channels (128 registers per lane); 8 waves of 4 channels (256) measured 25 % slower: too few waves to cover a phase
        if (dw_tall(p)) {
            DS_SET_MAX_LDS((dwconv7_mfma2_kernel<16, true>), M2<true>::LDS, "dwconv7_mfma2");
            hipLaunchKernelGGL((dwconv7_mfma2_kernel<16, true>), dim3(nb), dim3(1024), M2<true>::LDS, st, *p, g);
        } else {
            DS_SET_MAX_LDS((dwconv7_mfma2_kernel<16, false>), M2<false>::LDS, "dwconv7_mfma2");
            hipLaunchKernelGGL((dwconv7_mfma2_kernel<16, false>), dim3(nb), dim3(1024), M2<false>::LDS, st, *p, g);
        }
        DS_CHECK_LAUNCH("dwconv7_mfma2");
        return DS_OK;
    }
    if (const DwStrip g = dw_strip(p); g.on) {
        // bit 0: XCD-chunked item order (consecutive items = the channel blocks of one strip on ONE XCD: the 64-byte halves of an output line
        // two channel blocks share merge in that L2; +2 % on 8 of 9 layer shapes, same box), bit 1: strip fastest instead of channel block
        // fastest (no consistent gain); bits 2 - 4 (honoured by -DDS_DW_ABL builds only) are timing ablations with WRONG results (no output stores / no refill loads / one tap column
        // instead of seven) — r05 at C = 96, 256 x 64, batch 128: 555 us whole; arithmetic alone 347 (784 v_pk_fma_f32 per thread and tile:
        // ~12 k cycles per pair of co-resident tiles against 6.3 k of issue slots), stores alone 264, loads alone 157, none of them 111
#ifdef DS_DW_ABL          // diagnostic builds only (tools/build_variants.py abl=-DDS_DW_ABL=1): the product library ignores the ablation bits
        static const int order = getenv("DS_DW_STRIP_ORDER") ? atoi(getenv("DS_DW_STRIP_ORDER")) : 1;
#else
        static const int order = (getenv("DS_DW_STRIP_ORDER") ? atoi(getenv("DS_DW_STRIP_ORDER")) : 1) & 3;
#endif
        DS_SET_MAX_LDS(dwconv7_strip_kernel, ST_LDS, "dwconv7_strip");
        hipLaunchKernelGGL(dwconv7_strip_kernel, dim3(blocks * p->B), dim3(ST_NT), ST_LDS, st, *p, g.strips_w, g.ncblk, g.hparts, g.rows_per_part, order);
        DS_CHECK_LAUNCH("dwconv7_strip");
        return DS_OK;
    }
    if (dw_use_lds(p)) {
        const int nv = lt_nv(p), twl = lt_twl(p->W, nv), tw = 1 << twl, th = (2048 / nv) >> twl;
        const int tiles_w = (p->W + tw - 1) / tw, tiles_h = (p->H + th - 1) / th, ncblk = C / (nv * V);
        const size_t lds = (size_t)(tw + 6) * (th + 6) * nv * 16 + (size_t)49 * nv * V * sizeof(float) + 64;
        // (above 64 KB — the 8-wide tile of NV = 4 and both NV = 8 tiles — a kernel has to ask for its dynamic LDS)
#define DS_DW_LDS_LAUNCH(T_, TWL_, NV_)                                                                                                            \
        do {                                                                                                                                      \
            if (lds > 65536) DS_SET_MAX_LDS((dwconv7_lds_kernel<T_, TWL_, NV_>), lds, "dwconv7_lds");                                              \
            hipLaunchKernelGGL((dwconv7_lds_kernel<T_, TWL_, NV_>), dim3(blocks, p->B), dim3(LT_NT), lds, st, *p, tiles_w, tiles_w * tiles_h, ncblk); \
        } while (0)
        if (p->dtype == DS_BF16) {
            if (twl == 5) DS_DW_LDS_LAUNCH(bf16, 5, 4); else if (twl == 4) DS_DW_LDS_LAUNCH(bf16, 4, 4); else DS_DW_LDS_LAUNCH(bf16, 3, 4);
        } else if (nv == 8) {
            if (twl == 4) DS_DW_LDS_LAUNCH(float, 4, 8); else DS_DW_LDS_LAUNCH(float, 3, 8);
        } else {
            if (twl == 5) DS_DW_LDS_LAUNCH(float, 5, 4); else if (twl == 4) DS_DW_LDS_LAUNCH(float, 4, 4); else DS_DW_LDS_LAUNCH(float, 3, 4);
        }
#undef DS_DW_LDS_LAUNCH
        DS_CHECK_LAUNCH("dwconv7_lds");
        return DS_OK;
    }
    if (p->dtype == DS_BF16) hipLaunchKernelGGL(dwconv7_kernel<bf16>, dim3(blocks, p->B), dim3(DW_BLOCK), 0, st, *p, nstrip, CV);
    else hipLaunchKernelGGL(dwconv7_kernel<float>, dim3(blocks, p->B), dim3(DW_BLOCK), 0, st, *p, nstrip, CV);
    DS_CHECK_LAUNCH("dwconv7");
    return DS_OK;
}

#if DS_BOUNDS
extern "C" int ds_bounds_fetch_dwconv_gn(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

extern "C" int ds_pack_dw_weight(const float* w, int C, float* dst, void* stream) {
    DS_REQUIRE(w && dst && C > 0, "pack_dw: bad args");
    hipLaunchKernelGGL(pack_dw_kernel, dim3((49 * C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, C, dst);
    DS_CHECK_LAUNCH("pack_dw");
    return DS_OK;
}

extern "C" int ds_pack_dw_weight_mfma(const float* w, int C, void* dst, void* stream) {
    DS_REQUIRE(w && dst && C > 0, "pack_dw_mfma: bad args");
    const long total = (long)C * 6 * 64 * 8;
    hipLaunchKernelGGL(pack_dw_mfma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, C,
                       reinterpret_cast<bf16*>(dst));
    DS_CHECK_LAUNCH("pack_dw_mfma");
    return DS_OK;
}

extern "C" int ds_gn_finalize(const float* part, int B, int parts, double count, float eps, float* ab, void* stream) {
    DS_REQUIRE(part && ab && B > 0 && parts > 0 && count > 0, "gn_finalize: bad args");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), part, parts, count, eps, ab);
    DS_CHECK_LAUNCH("gn_finalize");
    return DS_OK;
}

extern "C" int ds_gn_stats(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* ab, void* stream) {
    DS_REQUIRE(x && ab && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "gn_stats: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DS_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16>, dim3(B * G), dim3(256), 0, st, (const bf16*)x, HW, C, G, eps, ab);
    else if (dtype == DS_F32) hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(B * G), dim3(256), 0, st, (const float*)x, HW, C, G, eps, ab);
    else DS_FAIL(DS_EINVAL, "gn_stats: dtype %d", dtype);
    DS_CHECK_LAUNCH("gn_stats");
    return DS_OK;
}

// blocks per sample of the streaming statistics pass (a function of the shape only)
static int gn_stream_blocks(int HW) {
    int n = (HW + 511) / 512;                    // >= 512 pixels per block (r04: 2048 left a 128 x 64 x 160 tensor at batch 64 with one block per CU)
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}
extern "C" size_t ds_gn_stats_ws_floats(int B, int HW, int C) { return (size_t)B * gn_stream_blocks(HW) * C * 2; }

extern "C" int ds_gn_stats_stream(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* ws, float* ab, void* stream) {
    DS_REQUIRE(x && ab && ws && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "gn_stats_stream: bad args");
    DS_REQUIRE(dtype == DS_F32 || dtype == DS_BF16, "gn_stats_stream: dtype %d", dtype);
    const int V = dtype == DS_BF16 ? 8 : 4, CV = C / V;
    DS_REQUIRE(C % V == 0 && CV <= 256 && ds_aligned16(x), "gn_stats_stream: C=%d must be a multiple of %d (at most %d) and x 16-byte aligned", C, V, 256 * V);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nblk = gn_stream_blocks(HW), ppb = (HW + nblk - 1) / nblk, threads = CV * (256 / CV);
    const size_t lds = (size_t)(threads / CV) * C * 2 * sizeof(float);
    if (dtype == DS_BF16) hipLaunchKernelGGL(gn_stats_stream_kernel<bf16>, dim3(nblk, B), dim3(threads), lds, st, (const bf16*)x, HW, C, ppb, ws);
    else hipLaunchKernelGGL(gn_stats_stream_kernel<float>, dim3(nblk, B), dim3(threads), lds, st, (const float*)x, HW, C, ppb, ws);
    DS_CHECK_LAUNCH("gn_stats_stream");
    hipLaunchKernelGGL(gn_stats_finish_kernel, dim3(B * G), dim3(64), 0, st, ws, nblk, C, G, (double)HW * (C / G), eps, ab);
    DS_CHECK_LAUNCH("gn_stats_finish");
    return DS_OK;
}

extern "C" int ds_gn_stats_finish(const float* ws, int B, int nblk, int C, int G, int HW, float eps, float* ab, void* stream) {
    DS_REQUIRE(ws && ab && B > 0 && nblk > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0, "gn_stats_finish: bad args");
    hipLaunchKernelGGL(gn_stats_finish_kernel, dim3(B * G), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ws, nblk, C, G, (double)HW * (C / G), eps, ab);
    DS_CHECK_LAUNCH("gn_stats_finish");
    return DS_OK;
}

extern "C" int ds_gn_apply(const ds_gn_apply_params* p, void* stream) {
    DS_REQUIRE(p && p->x && p->out && (p->gn_ab || p->gn_part) && p->gamma && p->beta, "gn_apply: null pointer");
    DS_REQUIRE(!p->gn_part || (p->G == 1 && !p->gn_ab && p->gn_parts > 0 && p->gn_count > 0), "gn_apply: partials need G == 1 and no gn_ab");
    DS_REQUIRE(p->dtype == DS_F32 || p->dtype == DS_BF16, "gn_apply: dtype %d", p->dtype);
    const int V = p->dtype == DS_BF16 ? 8 : 4;
    DS_REQUIRE(p->C % V == 0 && p->G > 0 && p->C % p->G == 0, "gn_apply: C=%d must be a multiple of %d and of G=%d", p->C, V, p->G);
    if (!ds_aligned16(p->x) || !ds_aligned16(p->out) || (p->res && !ds_aligned16(p->res)))
        DS_FAIL(DS_EALIGN, "gn_apply: pointers must be 16-byte aligned");
    const size_t nvec = (size_t)p->B * p->HW * (p->C / V);
    const int blocks = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if DS_BOUNDS
    {
        const long long es = V == 8 ? 2 : 4, n = (long long)p->B * p->HW * p->C * es;
        DsBxHost h(DS_K_GN_APPLY);
        h.set(DS_BX_SRC0, p->x, n).set(DS_BX_RES, p->res, n).set(DS_BX_OUT, p->out, n);
        h.set(DS_BX_GNPART, p->gn_part, (long long)p->B * p->gn_parts * 2 * 4);
        h.publish(st);
    }
#endif
    if (p->gn_part && p->dtype == DS_BF16 && p->act == DS_ACT_NONE && !p->cbias && 192 % (p->C / 8) == 0 && !getenv("DS_NO_GN_FAST")) {
        const int rows = 192 / (p->C / 8);
        int bx = (p->HW + rows * 4 - 1) / (rows * 4);
        const int cap = 4096 / p->B > 0 ? 4096 / p->B : 1;
        if (bx > cap) bx = cap;
        const int ppb = (p->HW + bx - 1) / bx;
        if (p->res) hipLaunchKernelGGL(gn_apply_lazy_fast_kernel<true>, dim3(bx, p->B), dim3(192), 0, st, *p, ppb);
        else hipLaunchKernelGGL(gn_apply_lazy_fast_kernel<false>, dim3(bx, p->B), dim3(192), 0, st, *p, ppb);
        DS_CHECK_LAUNCH("gn_apply_lazy_fast");
        return DS_OK;
    }
    if (p->gn_part) {
        const size_t per = (size_t)p->HW * (p->C / V);
        int bx = (int)((per + 255) / 256);
        const int cap = 2048 / p->B > 0 ? 2048 / p->B : 1;
        if (bx > cap) bx = cap;
        if (p->dtype == DS_BF16) hipLaunchKernelGGL(gn_apply_lazy_kernel<bf16>, dim3(bx, p->B), dim3(256), 0, st, *p);
        else hipLaunchKernelGGL(gn_apply_lazy_kernel<float>, dim3(bx, p->B), dim3(256), 0, st, *p);
        DS_CHECK_LAUNCH("gn_apply_lazy");
        return DS_OK;
    }
    // (every tier, every G since r04: the per-channel (scale, shift) table kernel — its SiLU is exp2f + v_rcp_f32, ~1 ulp each, where the
    // element-wise kernel below uses libm's expf and a true divide: the fp32 tier's bits differ from r03's by that much)
    static const bool no_table = getenv("DS_NO_GN_TABLE") != nullptr;
    if (p->gn_ab && p->C / V <= 256 && p->C * 8 <= 48 * 1024 && !no_table) {
        const int CV = p->C / V, threads = CV * (256 / CV), rows = threads / CV;
        int bx = (p->HW + rows * 8 - 1) / (rows * 8);                        // >= 8 pixels per thread
        const int cap = 8192 / p->B > 0 ? 8192 / p->B : 1;
        if (bx > cap) bx = cap;
        const int ppb = (p->HW + bx - 1) / bx;
        const size_t lds = (size_t)p->C * 2 * sizeof(float);
#define DS_GN_TAB(T_, A_) hipLaunchKernelGGL((gn_apply_table_kernel<T_, A_>), dim3(bx, p->B), dim3(threads), lds, st, *p, ppb)
#define DS_GN_TAB_T(T_)                                          \
    do {                                                         \
        if (p->act == DS_ACT_SILU) DS_GN_TAB(T_, DS_ACT_SILU);   \
        else if (p->act == DS_ACT_RELU) DS_GN_TAB(T_, DS_ACT_RELU); \
        else if (p->act == DS_ACT_GELU) DS_GN_TAB(T_, DS_ACT_GELU); \
        else DS_GN_TAB(T_, DS_ACT_NONE);                         \
    } while (0)
        if (p->dtype == DS_BF16) DS_GN_TAB_T(bf16);
        else DS_GN_TAB_T(float);
#undef DS_GN_TAB_T
#undef DS_GN_TAB
        DS_CHECK_LAUNCH("gn_apply_table");
        return DS_OK;
    }
    if (p->dtype == DS_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16>, dim3(blocks), dim3(256), 0, st, *p, nvec);
    else hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, *p, nvec);
    DS_CHECK_LAUNCH("gn_apply");
    return DS_OK;
}
