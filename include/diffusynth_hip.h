/*
 * diffusynth_hip.h — C ABI of libdiffusynth_hip.so (gfx950 / MI355X).
 *
 * The reference (WxuanYuan/diffusynth) is pure Python on PyTorch eager: it has no FFI, plugin or
 * operator registry.  Its boundary for the denoising hot path is the Python duck type
 *     eps = model(x, mapped_t, condition)           model/DiffSynthSampler.py:312,319
 * plus the nn.Module state-dict names of model/diffusion.py:21-175.  The entry points below are
 * therefore the ATen operator call sites of that path, one exported function per kernel class
 * (SURVEY.md §8a/§8b); each comment cites the reference line(s) the function replaces.  The Python
 * host code (diffusynth_amd/) mirrors ConditionedUnet / DiffSynthSampler on top of these and binds
 * them with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - activations between kernels are channels-last: [B][H][W][C], element type ds_dtype;
 *     the public boundary tensors (x, eps, latents) stay NCHW fp32 like the reference;
 *   - functions only enqueue work on `stream` (a hipStream_t passed as void*), never synchronise,
 *     never allocate, and are re-entrant (no global mutable state) => HIP-graph capturable;
 *   - return 0 on success or a negative DS_E* code; ds_last_error_string() (thread local) explains.
 */
#ifndef DIFFUSYNTH_HIP_H
#define DIFFUSYNTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_EINVAL (-1)   /* bad argument / unsupported shape */
#define DS_ELAUNCH (-2)  /* HIP launch error */
#define DS_EALIGN (-3)   /* pointer or channel count not aligned as the kernel requires */

typedef enum { DS_F32 = 0, DS_BF16 = 1 } ds_dtype;
typedef enum { DS_ACT_NONE = 0, DS_ACT_GELU = 1, DS_ACT_SILU = 2, DS_ACT_RELU = 3 } ds_act;

const char* ds_last_error_string(void);
int ds_abi_version(void);

/* ---------------------------------------------------------------- convolution (implicit GEMM on MFMA)
 * Replaces nn.Conv2d 3x3 p1 (diffusion_components.py:64,122,125; diffusion.py:174), 1x1
 * (components:128,93,180-181,263-264), Downsample Conv2d(4,2,1) (components:37-39), Upsample
 * ConvTranspose2d(4,2,1) (components:32-34, as 4 sub-pixel 2x2 convolutions), init Conv2d 7x7
 * (diffusion.py:82) and the VQGAN decoder convolutions (VQGAN.py:154-171,190-221,254-259,342).
 * Fused: zero-copy skip concat of two sources (components:236-249), GroupNorm(1,C) of the INPUT
 * folded into weights/epilogue (components:121,124,148), bias, GELU/SiLU/ReLU (components:123),
 * residual add (components:139,29) and per-sample sum / sum-of-squares partials for the next
 * GroupNorm. */
#define DS_CONV_TILE_128x192 0
#define DS_CONV_TILE_256x96 1
#define DS_CONV_TILE_128x32 2
#define DS_CONV_TILE_64x96 3          /* 64 x 192 block tile (name kept for ABI stability) */
/* (ids 4 .. 10 were the first- and second-generation LDS-halo kernels, retired in round 3: ds_conv_igemm rejects them) */
#define DS_CONV_TILE_HALO3_256x96 11     /* 3x3 stride-1 pad-1, bf16 (conv3x3_halo3.hip): 256-pixel patch x 96 channels per block (4 waves of 64 px x 96 ch,
                                            two blocks per CU), the input halo of a 32-channel chunk staged once in LDS (XOR-swizzled 64-byte rows),
                                            16x16x32 MFMAs, hand-scheduled K loop; wk_order = 1; fused res_conv, split precision (flags) and split-K
                                            (ksplit dividing Cin / 32, raw slices to slab + ds_conv_splitk_reduce) supported */
#define DS_CONV_TILE_QUAD_HALO3 12       /* Conv2d(4, 2, 1) and ConvTranspose2d(4, 2, 1) (Downsample / Upsample, components:88-93) on the
                                            HALO3 pipeline with four taps per chunk (conv_quad_halo3.hip): bf16, wk_order = 2, Cin % 32 == 0
                                            with (transposed ? 1 : 4) * Cin / 32 a multiple of 6; transposed: cout_pad = 4 * Cout, Cout % 96 == 0 */

/* ds_conv_params.flags (DS_CONV_TILE_HALO3_256x96): the split-precision tier runs a 3x3 convolution of fp32 tensors on bf16 matrix
 * cores as x*w ~ x_hi*w_hi + x_lo*w_hi + x_hi*w_lo (fp32 accumulate; x = x_hi + x_lo to 2^-17).
 *   SPLIT_IN : src0 = 2C bf16 channels per pixel (the hi plane, then the lo plane, of a C-channel fp32 tensor), C0 = 2C; wpk holds
 *              [W_hi | W_hi | W_lo] as 3C input channels (chunk-major, wk_order = 1; GroupNorm gain folded before the split)
 *   OUT_SPLIT: the result is stored as hi / lo bf16 planes (out_C = 2 * Cout bf16 channels): the SPLIT_IN format of the next layer
 *   OUT_F32  : the result (+ an fp32 residual `res`) is stored as fp32 (out_C counts fp32 elements) */
#define DS_CONV_F_SPLIT_IN 1
#define DS_CONV_F_OUT_SPLIT 2
#define DS_CONV_F_OUT_F32 4
#define DS_CONV_F_IN_F32 8               /* ds_conv1x1_x3 only: src0 / src1 are fp32 tensors, split into hi / lo bf16 inside the kernel */

#define DS_CONV_TILE_HALO3_N16 13        /* 3x3 stride 1 pad 1 with Cout <= 16 (the U-Net's final 96 -> 4 convolution; conv3x3_smalln.hip): bf16, wk_order = 1 with
                                            cout_pad = 16, Cin % 32 == 0, bias (+ GELU) epilogue only */

typedef struct {
    /* input: channels [0,C0) come from src0, [C0,C0+C1) from src1 placed at (off_h1,off_w1) */
    const void* src0; const void* src1;
    int32_t C0, C1, H, W, H1, W1, off_h1, off_w1;
    /* weights packed by ds_pack_conv_weight: [phase][kchunks][cout_pad][32] */
    const void* wpk;
    int32_t Cout, cout_pad, KH, KW, stride, pad_h, pad_w;
    int32_t Ho, Wo;              /* output positions iterated by the GEMM M dimension (per sample)   */
    int32_t transposed;          /* 1: 4-phase ConvTranspose2d(4,2,1); KH=KW=2, out is 2Ho x 2Wo     */
    /* output */
    void* out;                   /* NHWC, ds_dtype, out_C channels per pixel, written at out_c0.. in 16-B
                                    groups: Cout is rounded up to 4 (fp32) / 8 (bf16), pad channels = 0 */
    int32_t out_C, out_c0;
    int32_t out_nchw_f32;        /* must be 0 (reserved; use ds_nhwc_to_nchw for the fp32 NCHW boundary)  */
    /* epilogue */
    const float* bias;           /* [Cout] or NULL                                                    */
    const float* gn_ab;          /* [B][2] = (rstd, rstd*mean) of the input's GroupNorm(1,C), or NULL */
    const float* fold_t1; const float* fold_t2; /* [ncls][Cout]; t1 already contains the bias         */
    int32_t ncls;                /* 1 or 9 (3x3 pad 1 border classes)                                 */
    int32_t act;                 /* DS_ACT_NONE or DS_ACT_GELU, applied before the residual           */
    const void* res;             /* NHWC residual with out_C channels (same indexing as out) or NULL  */
    float* stats_part;           /* [B][gridDim.x*gridDim.y][2] partial (sum, sumsq) of stored values */
    int32_t B, dtype, tile;
    /* split-K (halo tiles only): ksplit > 1 makes ds_conv_igemm write raw fp32 partial sums of K-slice z to
     * slab[z][B][Ho*Wo][roundup(Cout,8)]; ds_conv_splitk_reduce then sums the slices and runs the epilogue. */
    int32_t ksplit;
    int32_t wk_order;            /* K order of wpk: 2 = quad tiles [chunk][2x2 tap][cout_pad][32] of DS_CONV_TILE_QUAD_HALO3 (transposed: chunk =
                                    Cin / 32 group, row n = phase * Cout + co holds w[ci][co][3 - py - 2a][3 - px - 2b]; strided: chunk =
                                    (parity plane, Cin / 32 group), row co holds w[co][ci][1 - p + 2a][1 - q + 2b]; packed on the host side by
                                    diffusynth_amd/engine.py:pack_quad_weights),
                                    0 = tap-major [tap*NCC + cc] (generic kernel),
                                    1 = chunk-major [cc*9 + tap] (DS_CONV_TILE_HALO3_256x96 / _N16: one pointer increment per step) */
    float* slab;
    /* alternative to gn_ab: the producer's raw (sum, sumsq) partials [B][gn_parts][2]; every wave reduces them
     * itself (float64) at kernel start, which removes the ds_gn_finalize launch between producer and consumer */
    const float* gn_part; int32_t gn_parts; float gn_eps; double gn_count;
    /* DS_CONV_TILE_HALO3_256x96 only: the ConvNeXt block's 1x1 res_conv (components:128,139) fused into this launch.
     * res_steps = (res_C0 + res_C1) / 32 > 0 (a multiple of 3) runs that many 32-channel K steps over the block input x
     * (two-source zero-copy concat like src0/src1 of the generic kernel, same H x W as the output) in front of the 3x3
     * K loop: acc = sum Wres[n][c] x[c] at the centre tap, divided by the GroupNorm factor a in registers, then the
     * nine-tap chunks accumulate on top and the epilogue's a * acc + shift yields res + a * conv3x3.  wpk must hold
     * res_steps tiles [cout_pad][32] (k_order 1 packing of the 1x1 weight) followed by the 3x3 tiles, res must be NULL,
     * res_bias[Cout] is added to every border class. */
    const void* res_src0; const void* res_src1;
    int32_t res_C0, res_C1, res_H1, res_W1, res_off_h1, res_off_w1;
    int32_t res_steps;
    int32_t flags;               /* DS_CONV_TILE_HALO3_256x96 only, 0 elsewhere: DS_CONV_F_* (split-precision tier) */
    const float* res_bias;
} ds_conv_params;

int ds_conv_igemm(const ds_conv_params* p, void* stream);
/* fp32 NHWC [npix][C] -> hi / lo bf16 planes [npix][2C] (the DS_CONV_F_SPLIT_IN format) for tensors produced by fp32 kernels */
int ds_split_planes(const float* x, void* out_bf16, long long npix, int C, void* stream);
/* 1x1 convolution (stride 1) of fp32 NHWC tensors in split precision on the bf16 matrix cores (tier "bf16x3": to_qkv, to_out, res_conv —
 * diffusion_components.py:128,139,263-264): flags = DS_CONV_F_IN_F32 | DS_CONV_F_OUT_F32, dtype DS_BF16, src0 / src1 fp32 with C0, C1
 * multiples of 32 (src1 placed at (off_h1, off_w1): pad_and_concat, components:210-249), wpk = [chunk of 32 input channels][hi, lo]
 * [cout_pad][32] bf16 with cout_pad a multiple of 96 (ds_conv1x1_x3_weight_elems elements; hi = bf16(w), lo = bf16(w - hi), PreNorm gain
 * folded before the split), out / res fp32 [B][H*W][out_C], bias or gn_ab + fold_t1 / fold_t2 (ncls = 1), optional stats_part
 * [B][ds_conv1x1_x3_stats_parts][2]. */
int ds_conv1x1_x3(const ds_conv_params* p, void* stream);
int ds_conv1x1_x3_stats_parts(const ds_conv_params* p);
size_t ds_conv1x1_x3_weight_elems(int Cin, int Cout);
int ds_conv_splitk_reduce(const ds_conv_params* p, void* stream);
/* number of (sum,sumsq) partial slots per sample that ds_conv_igemm writes for this problem */
int ds_conv_stats_parts(const ds_conv_params* p);
/* block N-tile of a tile id (weights must be packed with cout_pad = multiple of it) */
int ds_conv_tile_bn(int tile);

/* OIHW fp32 (or IOHW for transposed) -> packed kernel layout, optionally pre-scaled per input
 * channel by a GroupNorm gain (the GN fold).  dst holds nphase*kchunks*cout_pad*32 elements. */
typedef struct {
    const float* w;              /* [Cout][Cin][KH][KW], or [Cin][Cout][4][4] when transposed         */
    const float* gamma;          /* [Cin] or NULL                                                     */
    void* dst; int32_t dtype;
    int32_t Cout, Cin, cin_pad, KH, KW, cout_pad, transposed;
    int32_t k_order;             /* 0: k = tap*cin_pad + c (default); 1: chunk-major k = (c/32)*KH*KW*32 + tap*32 + c%32
                                    (cin_pad must be a multiple of 32; ds_conv_params.wk_order = 1)              */
} ds_pack_conv_params;
int ds_pack_conv_weight(const ds_pack_conv_params* p, void* stream);
size_t ds_pack_conv_elems(int Cin_pad, int KH, int KW, int cout_pad, int transposed);
/* fold tables for a GroupNorm(1,C)-fed convolution: t1[cls][o] = bias[o] + sum_{taps in cls,c} w*beta[c],
 * t2[cls][o] = sum w*gamma[c]; ncls = 9 for 3x3 pad 1, 1 for 1x1. */
int ds_conv_fold_tables(const float* w, const float* bias, const float* gamma, const float* beta,
                        int Cout, int Cin, int KH, int KW, float* t1, float* t2, void* stream);

/* ---------------------------------------------------------------- depthwise 7x7 (+bias +time bias)
 * Replaces ConvNextBlock.ds_conv and the broadcast add of mlp(time_emb) (components:118,131-136);
 * reads the skip concat from two sources; emits GroupNorm partials. */
typedef struct {
    const void* src0; const void* src1;
    int32_t C0, C1, H, W, H1, W1, off_h1, off_w1;
    const float* wt;             /* [49][C] tap-major fp32 (ds_pack_dw_weight)                        */
    const float* bias;           /* [C]                                                                */
    const float* tbias;          /* [B][tb_stride] time bias, this block's slice starts at tbias, or NULL */
    int32_t tb_stride;
    void* out;                   /* NHWC [B][H][W][C]                                                  */
    float* stats_part;           /* [B][parts][2]                                                      */
    int32_t B, dtype;
    /* bf16 only, optional: Toeplitz-expanded weights (ds_pack_dw_weight_mfma).  When given (and the channel
     * counts are multiples of 32) the 7x7 stencil runs on the matrix cores: per channel, a 16x16 output block
     * is A[16 rows][(dh, 24 cols)] x T_c[(dh, 24 cols)][16 cols] with T_c the banded matrix of the 7 row taps. */
    const void* wexp;
    int32_t out_split;           /* fp32 only: store the result as two bf16 planes (hi = bf16(v), then lo = bf16(v - hi)) of a 2C-channel image,
                                    the DS_CONV_F_SPLIT_IN input format of the split-precision 3x3 convolution */
    int32_t strip;               /* fp32 32-channel blocks only: 0 = the library chooses between the tile kernel and the strip kernel (an LDS ring walking
                                    down a 16-column strip: every input row leaves HBM once) by shape and batch, 1 = the strip kernel wherever its shape
                                    conditions hold (W >= 16, H >= 32), 2 = never.  Outputs are identical bit for bit; the statistics partials are
                                    grouped differently (ds_dwconv_stats_parts answers for the same setting) */
} ds_dwconv_params;
int ds_dwconv7(const ds_dwconv_params* p, void* stream);
int ds_dwconv_stats_parts(const ds_dwconv_params* p);
int ds_pack_dw_weight(const float* w_c1kk, int C, float* dst_tap_major, void* stream);
/* [C][1][7][7] fp32 -> [C][6 k-steps][64 lanes][8] bf16 B-operand fragments of v_mfma_f32_16x16x32_bf16 */
int ds_pack_dw_weight_mfma(const float* w_c1kk, int C, void* dst_bf16, void* stream);

/* ---------------------------------------------------------------- GroupNorm
 * nn.GroupNorm(1,C) (components:121,124,148,181,264): statistics are produced as partials by the
 * producing kernel; ds_gn_finalize reduces them (float64) to (rstd, rstd*mean) per sample.
 * ds_gn_apply is the explicit normalise+affine(+act)(+residual) pass used where the norm cannot be
 * folded into a following convolution (attention output norm + Residual, components:181,29; VQGAN
 * Normalize + swish/ReLU, VQGAN.py:12-27,225-226,360-361).  ds_gn_stats computes per-(sample,group)
 * statistics of a tensor directly (groups > 1: components:65, VQGAN.py:17). */
int ds_gn_finalize(const float* stats_part, int B, int parts, double count, float eps, float* gn_ab, void* stream);
int ds_gn_stats(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* gn_ab, void* stream);
/* the same statistics as one streaming pass (16-byte loads over whole pixels, per-channel block partials in `ws`, channels
 * folded into groups in float64 by a second tiny launch): what the engines use; C must be a multiple of 4 (fp32) / 8 (bf16). */
size_t ds_gn_stats_ws_floats(int B, int HW, int C);
int ds_gn_stats_stream(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* ws, float* gn_ab, void* stream);
typedef struct {
    const void* x; const void* res; void* out;   /* NHWC [B][HW][C]; res may be NULL                  */
    const float* gn_ab;          /* [B][G][2]                                                          */
    const float* gamma; const float* beta;       /* [C]                                                */
    const float* cbias;          /* [B][cb_stride] per-(sample,channel) add after norm+act (components:98-101), or NULL    */
    int32_t cb_stride;
    int32_t B, HW, C, G, act, dtype;
    /* G == 1 only: raw (sum, sumsq) partials of x instead of gn_ab (gn_ab NULL) — the kernel reduces them itself, like
     * ds_conv_params.gn_part, which saves the ds_gn_finalize launch */
    const float* gn_part; int32_t gn_parts; float gn_eps; double gn_count;
} ds_gn_apply_params;
int ds_gn_apply(const ds_gn_apply_params* p, void* stream);

/* ---------------------------------------------------------------- linear attention
 * LinearCrossAttentionAdd.forward / LinearCrossAttention.forward (components:271-293,187-207) and
 * VQGAN LinearAttention (VQGAN.py:261-272) on a precomputed qkv tensor [B][N][3*heads*32]:
 *   pass 1: per (sample, head, segment) softmax_n(k) . v^T partial context (max, sum, 32x32)
 *   combine: merges segments (+ the optional extra key/value token of linear_cat)
 *   pass 2: q~ = softmax_d(q + label_q) * scale;  out[n][h*32+e] = sum_d ctx[d][e] q~[d]      */
typedef struct {
    const void* qkv;             /* [B][N][3*heads*32]: q | k | v, each (head, d)                      */
    int32_t B, N, heads, dtype;
    int32_t nseg;                /* segments of N used by pass 1                                       */
    float* part;                 /* [B][heads][nseg][32+32+1024] scratch                               */
    float* ctx;                  /* [B][heads][32][32]                                                 */
    const float* label_q;        /* [B][lq_stride]: + (head*32+d); NULL = none                         */
    const float* label_k; const float* label_v;  /* linear_cat extra token, [B][l*_stride]; NULL = none */
    int32_t lq_stride, lk_stride, lv_stride;
    int32_t q_softmax;           /* 1: softmax over d and * scale (U-Net); 0: raw q (VQGAN)            */
    float scale;
    void* out;                   /* [B][N][heads*32]                                                   */
} ds_attn_params;
int ds_linattn_context(const ds_attn_params* p, void* stream);   /* pass 1 + combine */
int ds_linattn_output(const ds_attn_params* p, void* stream);    /* pass 2           */
size_t ds_linattn_part_floats(int B, int heads, int nseg);

/* Fused form of the whole Residual(PreNorm(LinearCrossAttentionAdd)) block up to the output GroupNorm
 * (components:142-152,252-293), bf16, heads = 4 x 32, C in {96,192,384}: x is the only activation stream
 * (no qkv tensor).  context = k/v projection + softmax_n + k.v^T (+ segment combine); output = q projection
 * + softmax_d + ctx^T.q + to_out 1x1 + bias + GroupNorm partials.  Follow with ds_gn_finalize + ds_gn_apply(res = x). */
typedef struct {
    const void* x;               /* [B][N][C] bf16                                                     */
    int32_t B, N, C, nseg;
    const void* wqkv;            /* [384][C] bf16 row-major, PreNorm gain folded in (ds_pack_attn_fused) */
    const float* t1; const float* t2;   /* [384] fold tables of the PreNorm (ds_conv_fold_tables, bias NULL) */
    const float* gn_ab;          /* [B][2] statistics of x, or NULL when gn_part is given               */
    const float* gn_part; int32_t gn_parts; float gn_eps; double gn_count;   /* raw partials of x (see ds_conv_params) */
    const float* label_q;        /* [B][lq_stride] or NULL                                              */
    int32_t lq_stride; float scale;
    float* part; float* ctx;     /* scratch as in ds_attn_params (heads = 4)                            */
    const void* wout_perm;       /* [C][128] bf16, columns permuted per head (ds_pack_attn_fused)       */
    const float* bias_out;       /* [C]                                                                 */
    void* y;                     /* [B][N][C] bf16 = to_out.0 output                                    */
    float* stats_part;           /* [B][ds_attn_fused_stats_parts][2]                                   */
    void* mfold;                 /* [B][C][128] bf16 scratch or NULL.  Given (C = 96 / 192): the output pass runs attn_out2 — to_out folded
                                    into the per-sample context (M_b = Wout . ctx_b^T), wave = pixel tile, no LDS exchange (attn_out2.hpp);
                                    NULL: the first-generation output kernel.  ds_attn_fused_stats_parts depends on it.               */
    int32_t gen;                 /* kernel generation: 0 = chosen by the batch (the second-generation kernels — wave = 32-pixel tile, weights in
                                    LDS, attn_out2.hpp — pay from about 100 samples on: context pass from B >= 96, output pass from B >= 32 at
                                    C = 96 and B >= 96 at C = 192; below that a block stages 50 - 100 KB of weights for a handful of tiles);
                                    1 / 2 = the first / second generation wherever it exists (tests, A/B)                              */
} ds_attn_fused_params;
int ds_pack_attn_fused(const float* wqkv_384xC, const float* gamma_C, const float* wout_Cx128, void* wqkv_bf16,
                       void* wout_perm_bf16, int C, void* stream);
int ds_attn_fused_context(const ds_attn_fused_params* p, void* stream);
int ds_attn_fused_output(const ds_attn_fused_params* p, void* stream);
int ds_attn_fused_stats_parts(const ds_attn_fused_params* p);
/* segments of partials per (sample, head) the context pass wants for gen = 0 at this shape (nseg; part = ds_linattn_part_floats(B, 4, nseg)):
 * the second generation works one segment per wave and wants one round of blocks (2048 / B, 1024 / B at C = 384), the first N / 128 <= 32 */
int ds_attn_fused_segments(int B, int N, int C);
/* the same for an explicit kernel generation (ds_attn_fused_params.gen = 1 / 2; 0 = by batch, as above): a caller that forces a generation
 * sizes `part` from THIS count */
int ds_attn_fused_segments_gen(int B, int N, int C, int gen);

/* ---------------------------------------------------------------- fused LinearAttention of the VQGAN (bf16 tier; csrc/vq_attn.hip)
 * VQGAN.py:246-272 (heads = 1, dim_head = 32, softmax over n on k only): q enters linearly, so after the context the block is one 1x1
 * convolution with a per-sample weight, y = (Wnin + Wout ctx_b^T Wq) x + bias.  Replaces to_qkv + ds_linattn_context + ds_linattn_output +
 * to_out / nin_shortcut of the decoder / encoder plans — no qkv tensor; x is read twice, y written once.  C in {80, 160}.
 *   ds_vq_attn_context: ctx[b][d][e] (softmax_n(Wk x) (Wv x)^T, segment partials merged);
 *   ds_vq_attn_output:  W_b per sample (fp32 fold, bf16 operands), y = W_b x + bias, optional per-channel statistics of y. */
typedef struct {
    const void* x;               /* [B][N][C] bf16                                                      */
    int32_t B, N, C, nseg;       /* nseg = ds_vq_attn_segments(B, N, C): 4 per block (one partial per wave) */
    const void* wqkv;            /* [96][C] bf16 = to_qkv.weight (rows q | k | v)                       */
    const float* wq;             /* [32][C] fp32 = to_qkv.weight[:32]                                   */
    const float* wout;           /* [C][32] fp32 = to_out.weight                                        */
    const float* wnin;           /* [C][C] fp32 = nin_shortcut.weight, or NULL (with_skip = False)      */
    const float* bias;           /* [C] = to_out.bias (+ nin_shortcut.bias)                             */
    float* part; float* ctx;     /* scratch: ds_linattn_part_floats(B, 1, nseg) floats / B * 1024 floats */
    void* wfold;                 /* scratch: ds_vq_attn_wfold_bytes(B, C)                               */
    void* y;                     /* [B][N][C] bf16                                                      */
    float* stats_ws;             /* optional [B][nseg / 4][C][2]: per-channel (sum, sum of squares) of the stored y per block —
                                    ds_gn_stats_finish(slots = nseg / 4) turns them into the next Normalize's statistics */
} ds_vq_attn_params;
int ds_vq_attn_segments(int B, int N, int C);
size_t ds_vq_attn_wfold_bytes(int B, int C);
int ds_vq_attn_context(const ds_vq_attn_params* p, void* stream);
int ds_vq_attn_output(const ds_vq_attn_params* p, void* stream);

/* ---------------------------------------------------------------- fused linear attention, split precision (tier "bf16x3")
 * The same block (Residual(PreNorm(LinearCrossAttentionAdd)) up to the output GroupNorm, components:142-152,252-293) on FP32 tensors,
 * every dense product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the bf16 matrix cores with fp32 accumulation (csrc/attn_x3.hip): replaces
 * to_qkv (ds_conv1x1_x3) + ds_linattn_context + ds_linattn_output + to_out of that tier — no qkv tensor.
 *   ds_attn_x3_context: k / v / q projections of x (read once from HBM), softmax_n(k), ctx = k v^T per (sample, head) (+ segment combine),
 *                       softmax_d(q) * scale written as ready-made MFMA operands ("q planes");
 *   ds_attn_x3_output:  M_b = Wout ctx_b^T per sample, y = M_b q~ + bias (fp32 NHWC) + GroupNorm partials of y (form A: follow with
 *                       ds_gn_apply(res = x), whose gn_part form reduces stats_part itself), or the finished block output (form B, `out`).
 * heads = 4 x 32, C in {96, 192, 384}. */
typedef struct {
    const float* x;              /* [B][N][C] fp32                                                      */
    int32_t B, N, C, nseg;       /* nseg: ds_attn_x3_segments(B, N, C) (one segment of partials per wave) */
    const void* wqkv_hl;         /* [2][384][C] bf16: hi plane, lo plane of to_qkv.weight * PreNorm gain (ds_pack_attn_x3) */
    const float* t1; const float* t2;   /* [384] fold tables of the PreNorm (ds_conv_fold_tables, bias NULL) */
    const float* gn_ab;          /* [B][2] statistics of x, or NULL when gn_part is given               */
    const float* gn_part; int32_t gn_parts; float gn_eps; double gn_count;   /* raw partials of x (see ds_conv_params) */
    const float* label_q;        /* [B][lq_stride] or NULL                                              */
    int32_t lq_stride; float scale;
    float* part; float* ctx;     /* scratch as in ds_attn_params (heads = 4): ds_linattn_part_floats(B, 4, nseg) / B*4*1024 floats */
    void* qplanes;               /* scratch, ds_attn_x3_qplane_bytes(B, N)                              */
    void* mfold;                 /* scratch, ds_attn_x3_mfold_bytes(B, C)                               */
    const float* wout;           /* [C][128] fp32 = to_out.0.weight                                     */
    const float* bias_out;       /* [C]                                                                 */
    float* y;                    /* [B][N][C] fp32 = to_out.0 output (form A), or NULL with `out` (form B)   */
    float* stats_part;           /* [B][ds_attn_x3_stats_parts][2] (form A: may be NULL; form B: required scratch) */
    /* form B — the whole Residual(PreNorm(attention)) block: out = x + GroupNorm(1, C)(y) * on_gamma + on_beta (to_out.1, components:264 /
     * :22-29) without materialising y.  ds_attn_x3_output then runs pass 2 TWICE: once for the statistics of y alone, once more to
     * normalise it and add the residual while it is computed — two matrix passes instead of a 3 C N x 4 B apply pass over HBM.   */
    float* out;                  /* [B][N][C] fp32 or NULL                                              */
    const float* on_gamma; const float* on_beta;   /* [C] affine of the output GroupNorm (to_out.1)     */
    float on_eps;
    void* out_planes;            /* form B: the block output also (or, with out = NULL, only) as hi / lo bf16 planes [B][N][2C] — the input
                                    format of a DS_CONV_F_SPLIT_IN convolution (the Down / Upsample that follows): no ds_split_planes pass */
} ds_attn_x3_params;
int ds_pack_attn_x3(const float* wqkv_384xC, const float* gamma_C, void* wqkv_hl, int C, void* stream);
int ds_attn_x3_context(const ds_attn_x3_params* p, void* stream);
int ds_attn_x3_output(const ds_attn_x3_params* p, void* stream);
int ds_attn_x3_stats_parts(const ds_attn_x3_params* p);
int ds_attn_x3_segments(int B, int N, int C);
size_t ds_attn_x3_qplane_bytes(int B, int N);
size_t ds_attn_x3_mfold_bytes(int B, int C);

/* ---------------------------------------------------------------- diagnostics
 * libdiffusynth_hip_bounds.so (tools/build_variants.py bounds; -DDS_BOUNDS=1) checks every global access of the
 * convolution / depthwise / attention / GroupNorm-apply kernels against the operand extents implied by the parameter
 * structs and records the first violation per translation unit instead of performing it.  Returns the number of
 * records (0 = every access in range) and a description in buf; -1 from the product build. */
int ds_bounds_report(char* buf, int n, int reset);

/* ---------------------------------------------------------------- few-output fp32 3x3 (fp32 / bf16x3 tiers)
 * The U-Net's final_conv = Conv2d(dim, out_dim = 4, 3, padding = 1) (diffusion.py:103-105) on fp32 NHWC tensors: out [B][H][W][4] fp32 =
 * conv3x3(x [B][H][W][C], stride 1, zero padding 1) + bias, outputs beyond Cout zero.  C % 32 == 0, Cout <= 4.  ds_pack_conv3x3_f32_n4 turns
 * the Conv2d-layout weight [Cout][C][3][3] (+ bias or NULL) into ds_conv3x3_f32_n4_weight_floats(C) floats. */
size_t ds_conv3x3_f32_n4_weight_floats(int C);
int ds_pack_conv3x3_f32_n4(const float* w, const float* bias, int Cout, int C, float* dst, void* stream);
int ds_conv3x3_f32_n4(const float* x, int B, int H, int W, int C, const float* wpk, float* out, void* stream);

/* ---------------------------------------------------------------- conditioning MLPs
 * SinusoidalPositionEmbeddings (components:42-56), nn.Linear / GELU stacks (diffusion.py:99-105,
 * components:112-116,155-168,267-268).  y[b][o] = bias[o] + sum_k act_in(x[b][k]) W[o][k], fp32. */
int ds_sinusoid(const int64_t* t, const float* freqs, int B, int half, float* out, void* stream);
int ds_linear(const float* x, int x_stride, const float* W, const float* bias, int B, int K, int O, int act_in,
              float* y, int y_stride, void* stream);
/* y[i] = act(x[i]), fp32, in place allowed (act in {DS_ACT_NONE, DS_ACT_GELU, DS_ACT_SILU}): the activation in front of a wide nn.Linear
 * stack over one small input (the `mlp = Sequential(GELU, Linear)` of all 44 blocks, components:112-116) applied ONCE instead of once per
 * 16 outputs by ds_linear's act_in — the same fp32 function, so the results are identical. */
int ds_activation(const float* x, size_t n, int act, float* y, void* stream);

/* out[b][:] = LayerNorm(a[b][:] + r[b][:]) * gamma + beta over D (biased variance, eps inside the sqrt), fp32:
 * the tail of ProjectionLayer.forward (multimodal_model.py:29-31; SURVEY 8f row 3, text-condition head). */
int ds_add_layernorm(const float* a, const float* r, const float* gamma, const float* beta, int B, int D, float eps,
                     float* out, void* stream);

/* ---------------------------------------------------------------- layout converts at the boundary */
int ds_nchw_to_nhwc(const float* x, int B, int C, int H, int W, void* out, int C_pad, int dtype, void* stream);
/* The VQGAN decoder's first layer on the NCHW latent (VQGAN.py:345, Conv2d(embedding_dim, hidden, 1)): layout change + 1x1 convolution in one
 * pass, fp32 NCHW in, bf16 NHWC [B][HW][Cout] out.  Cin in {4, 8}, Cout % 8 == 0; w = [Cout][Cin] fp32, bias [Cout] or NULL. */
int ds_conv1x1_in_nchw(const float* x_nchw, int B, int Cin, int HW, const float* w, const float* bias, int Cout, void* out, void* stream);
/* dst[0, nbytes) = dst[nbytes, 2 nbytes) = src[0, nbytes): the two halves of a classifier-free-guidance batch (DiffSynthSampler.py:311-320 evaluates
 * model(cat([x, x]), cat([t, t]), cat([uncond, cond]))) are identical up to the first operator that reads the condition — the plan computes
 * that prefix once at half the batch and duplicates its result (engine.py, `paired`).  dst == src: the first half is in place already,
 * only dst[nbytes, 2 nbytes) is written (r05: the plan computes the prefix into the first half of the full-batch tensor). */
int ds_dup_batch(const void* src, void* dst, size_t nbytes, void* stream);
int ds_nhwc_to_nchw(const void* x, int dtype, int B, int C, int C_stride, int H, int W, float* out, void* stream);

/* ---------------------------------------------------------------- sampler step
 * DiffSynthSampler.ddim_sample (DiffSynthSampler.py:311-343): classifier-free-guidance combine,
 * predicted x0, sigma, direction and the update — and, for inpainting, the q_sample + mask blend of
 * p_sample_loop (DiffSynthSampler.py:499-510, 271-294).  fp32 NCHW, reference operation order,
 * no fused multiply-add: bit-exact with the CPU reference given identical inputs. */
typedef struct {
    const float* x; const float* eps; const float* eps_cond; /* eps_cond != NULL => eps is the uncond half */
    const float* noise; float* out;
    const float* coef;           /* [B][5]: sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma */
    float cfg_scale;
    /* inpaint blend (mode 0 none, 1: mask*q_sample(guide)+(1-mask)*x', 2: mask*guide+(1-mask)*x') */
    int32_t blend_mode;
    const float* guide; const float* init_noise; const float* mask; /* mask [B][1][H][W] (mask_chw == 0)  */
    const float* qcoef;          /* [B][2]: sqrt(acp[t-1]), sqrt(1-acp[t-1])                           */
    int32_t B, CHW, HW;
    int32_t mask_chw;            /* != 0: mask is [B][C][H][W] (the reference blends by broadcasting, and its
                                    inpaint UI passes a per-channel mask: inpaint_with_text.py:229-231)    */
} ds_step_params;
int ds_ddim_step(const ds_step_params* p, void* stream);
/* counter-based N(0,1) generator (Philox4x32-10 + Box-Muller) for the throughput mode */
int ds_philox_normal(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
/* column gather of the "repeat" noise layout (DiffSynthSampler.py:97-167): out[b][c][h][j] = src[b][c][h][cols[j]] */
int ds_gather_cols(const float* src, int rows, int src_w, const int32_t* cols, int out_w, float* out, void* stream);

/* ---------------------------------------------------------------- VQ + vocoder tail
 * VectorQuantizerEMA.forward eval (VQGAN.py:98-146): nearest code (first minimum, as torch.argmin); writes quantised NCHW fp32 (the
 * straight-through form z + (e - z)) and int64 indices.  Up to 8192 codes of dimension 4 run on the matrix cores: |e|^2 - 2 z.e in split
 * precision (three bf16 parts per fp32 number) — the codebook is re-packed into a per-device module buffer on every call, so concurrent calls
 * on different streams of one device must not overlap; larger codebooks (and DS_VQ_SCALAR=1) take the scalar kernel with the reference's
 * |z|^2 + |e|^2 - 2 z.e expression. */
int ds_vq_nearest(const float* z_nchw, const float* codebook, const float* code_sqnorm, int B, int D, int HW,
                  int ncodes, float* q_nchw, int64_t* idx, void* stream);
/* The quantiser's scalars from its outputs (VQGAN.py:62-73 / :131-144): out3[0] = mean((q - z)^2), out3[1] = perplexity =
 * exp(-sum_j p_j log(p_j + 1e-10)) (p = usage frequencies of the codes), out3[2] = the module's loss: commitment_cost * out3[0] (ema != 0:
 * VectorQuantizerEMA) or out3[0] + commitment_cost * out3[0] (VectorQuantizer).  ws: ds_vq_stats_ws_bytes. */
size_t ds_vq_stats_ws_bytes(int ncodes);
int ds_vq_stats(const float* z_nchw, const float* q_nchw, const int64_t* idx, int B, int D, int HW, int ncodes, float commitment_cost, int ema,
                float* out3, void* ws, void* stream);
/* Decoder tail activations (VQGAN.py:394-398): softplus / tanh / tanh on NHWC[.,C_stride] -> NCHW fp32 [B][3][H][W] */
int ds_decoder_tail(const void* x, int dtype, int B, int C_stride, int HW, float* out, void* stream);
/* The body of the VQGAN decoder's 80-channel ResnetBlock as one kernel (csrc/conv3x3_c80.hip; VQGAN.py:223-244 with temb = None, no
 * nin_shortcut): out = [x +] conv3x3(act(GroupNorm(G, 80)(x))) + bias, bf16 NHWC.  wpk = ds_conv3x3_c80_weight_elems() bf16 written by
 * ds_pack_conv3x3_c80 from the fp32 [80][80][3][3] Conv2d weight; gn_ab [B][G][2] (rstd, rstd * mean) with gamma / beta [80] and
 * act in {DS_ACT_NONE, DS_ACT_RELU, DS_ACT_SILU}: applied to x on load (NULL: the convolution reads x as it is); add_x: add the raw x
 * (the block's residual).  out must not alias x.  A sample must stay below 256 MB. */
size_t ds_conv3x3_c80_weight_elems(void);
int ds_pack_conv3x3_c80(const float* w, int Cout, int Cin, void* dst, void* stream);
int ds_conv3x3_c80(const void* x, int B, int H, int W, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                   const float* gamma, const float* beta, int act, int add_x, float* stats_ws, void* stream);
/* stats_ws (both 80-channel kernels; NULL: none): [B][slots][80][2] floats, slots = ds_*_c80_stats_slots(B, H, W) — per-channel (sum, sum of
 * squares) partials of the OUTPUT, every slot written; ds_gn_stats_finish(stats_ws, B, slots, 80, G, output pixels per sample, eps, ab) turns
 * them into the (rstd, rstd * mean) pairs of GroupNorm(G, 80) of that output: the next Normalize needs no pass over the tensor. */
int ds_conv3x3_c80_stats_slots(int B, int H, int W);
/* The same block as two launches: h = act(GroupNorm(res)) written by ds_gn_apply, then out = res + conv3x3(h) + bias (no arithmetic while the halo
 * is staged: 377 us against 637 us at 64 x 256 x 128 x 80, + ~110 us for the apply pass). */
int ds_conv3x3_c80_res(const void* h, const void* res, int B, int H, int W, const void* wpk, const float* bias, void* out, float* stats_ws,
                       void* stream);
int ds_convt4x4_c80_stats_slots(int B, int H, int W, int Cin);
/* second stage of ds_gn_stats_stream on its own: ws [B][nblk][C][2] per-channel partial sums -> ab [B][G][2] */
int ds_gn_stats_finish(const float* ws, int B, int nblk, int C, int G, int HW, float eps, float* ab, void* stream);

/* ConvTranspose2d(Cin, 80, 4, 2, 1) with Cin = 80 or 160, bf16 NHWC in / out, on its own kernel (csrc/convt4x4_c80.hip): the VQGAN decoder's
 * two Upsample layers (VQGAN.py Decoder `up` layers, SURVEY §8a tail row).  x [B][H][W][Cin]; wpk = ds_convt4x4_c80_weight_elems(Cin) bf16
 * written by ds_pack_convt4x4_c80 from the fp32 [Cin][80][4][4] weight (ConvTranspose2d layout [Cin][Cout][kh][kw]); bias [80] or NULL;
 * out [B][2H][2W][80] bf16.  gn_ab [B][G][2] (rstd, rstd * mean per group, e.g. from ds_gn_stats_stream) with gamma / beta [Cin]: the
 * layer reads relu(GroupNorm(G, Cin)(x)) instead of x (the decoder's Normalize + ReLU), applied on load; NULL: plain x.
 * stats_ws: see ds_conv3x3_c80.  A sample must stay below 256 MB. */
size_t ds_convt4x4_c80_weight_elems(int Cin);
int ds_pack_convt4x4_c80(const float* w, int Cin, int Cout, void* dst, void* stream);
int ds_convt4x4_c80(const void* x, int B, int H, int W, int Cin, const void* wpk, const float* bias, void* out, const float* gn_ab, int G,
                    const float* gamma, const float* beta, float* stats_ws, void* stream);

/* The U-Net's 7x7 init convolution (<= 4 real input channels -> 96, stride 1, pad 3; bf16 NHWC in / out) on its own kernel
 * (csrc/conv7x7_c4.hip).  Replaces: ConditionedUnet.init_conv = nn.Conv2d(channels, init_dim, 7, padding=3), model/DiffSynth.py
 * (SURVEY §8a row "init_conv").  x [B][H][W][Cx] with Cx = 8 (the engine's padded input image; channels 4..7 ignored) or 4;
 * wpk = ds_conv7x7_c4_weight_elems() bf16 written by ds_pack_conv7x7_c4 from the fp32 [96][Cin][7][7] weight; bias [96] or NULL;
 * out [B][H][W][96] bf16.  A sample must stay below 256 MB. */
size_t ds_conv7x7_c4_weight_elems(void);
int ds_pack_conv7x7_c4(const float* w, int Cout, int Cin, void* dst, void* stream);
int ds_conv7x7_c4(const void* x, int B, int H, int W, int Cx, const void* wpk, const float* bias, void* out, void* stream);
/* The same convolution in split precision for the bf16x3 tier: x [B][H][W][4] FP32, out [B][H][W][96] FP32 = x_hi w_hi + x_lo w_hi + x_hi w_lo
 * on bf16 MFMAs with fp32 accumulation (x split on its way to LDS); wpk = 2 x ds_conv7x7_c4_weight_elems() bf16 (hi fragments, then lo)
 * written by ds_pack_conv7x7_c4_x3. */
int ds_pack_conv7x7_c4_x3(const float* w, int Cout, int Cin, void* dst, void* stream);
int ds_conv7x7_c4_x3(const float* x, int B, int H, int W, const void* wpk, const float* bias, float* out, void* stream);

/* The decoder's last ResnetBlock(C -> 3) and the output activations in one pass over its input (VQGAN.py:177-244,390-398), bf16:
 * out[b] = [softplus, tanh, tanh](conv3x3(swish(GroupNorm(G, C)(x))) + nin_shortcut_1x1(x)).  x [B][H][W][C] bf16 (C % 8 == 0, C <= 96, W > 8),
 * gn_ab [B][G][2] = (rstd, rstd * mean) of x (ds_gn_stats / ds_gn_stats_stream), gamma / beta [C], w3 = the 3x3 weight packed by
 * ds_pack_conv_weight with cin_pad 96, cout_pad 16, k_order 1 (bf16), b3 [3], wnin [3][C] fp32, bnin [3]; out [B][3][H][W] fp32. */
int ds_dec_final(const void* x, int B, int H, int W, int C, const float* gn_ab, int G, const float* gamma, const float* beta,
                 const void* w3, const float* b3, const float* wnin, const float* bnin, float* out, void* stream);
/* ISTFT+ and iSTFT (tools.py:334-345,185-191; librosa.istft(D, hop_length=256, win_length=1024) at
 * webUI/natural_language_guided_4/utils.py:241): enc [B][3][F][T] fp32 -> audio [B][hop*(T-1)] fp32.
 * n_fft = 2*F, window = periodic Hann(n_fft), center=True. */
int ds_istft_plus(const float* enc, int B, int F, int T, int hop, float* frames_ws, float* audio, void* stream);
size_t ds_istft_ws_floats(int B, int F, int T);
/* Audio -> STFT+ representation: librosa.stft(y, n_fft=1024, hop_length=hop, win_length=1024)
 * (call sites load_presets.py:68, sound2sound_with_text.py:85, inpaint_with_text.py:97) + tools.pad_STFT
 * (tools.py:170-182: drop the DC row, zero-pad time to T_out) + tools.encode_stft (tools.py:320-331), fused:
 * audio [B][L] fp32 -> enc [B][3][512][T_out] fp32, T_out >= 1 + L/hop.  reflect_pad: 0 = zero padding of the
 * centred frames (librosa >= 0.10 default), 1 = reflect (older librosa). */
int ds_stft_plus(const float* audio, int B, int L, int hop, int reflect_pad, int T_out, float* enc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
