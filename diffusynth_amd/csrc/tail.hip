// Latent -> audio tail: nearest-code vector quantiser, decoder output activations, ISTFT+ and the
// inverse STFT (per-frame 1024-point inverse FFT in LDS, windowed overlap-add as a gather) (gfx950).
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ VQ
// distance in the reference's form |z|^2 + |e|^2 - 2 z.e ; first minimum wins (torch.argmin).
// The code index is loop-uniform, so the codebook row is fetched through the scalar cache and
// broadcast: no LDS needed, each thread owns one latent pixel.
template <int D>
__global__ __launch_bounds__(256) void vq_kernel(const float* z, const float* cb, const float* esq, int HW, int ncodes, float* q,
                                                 int64_t* idx, size_t npix) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t b = i / HW, pix = i % HW;
    float zv[D];
    float zz = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        zv[c] = z[(b * D + c) * HW + pix];
        zz += zv[c] * zv[c];
    }
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < ncodes; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) dot = fmaf(zv[c], cb[(size_t)j * D + c], dot);
        const float dist = (zz + esq[j]) - 2.0f * dot;
        if (dist < best) {
            best = dist;
            bi = j;
        }
    }
    if (idx) idx[i] = bi;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float e = cb[(size_t)bi * D + c];
        q[(b * D + c) * HW + pix] = zv[c] + (e - zv[c]);  // straight-through form of VQGAN.py:140
    }
}

// ---- the same search on the matrix cores (r03).  d_j = |e_j|^2 - 2 z.e_j (|z|^2 is the same for every code) is ONE v_mfma_f32_16x16x32_bf16 per
// 16 pixels x 16 codes: the 32 K slots carry the dot product in split precision — every fp32 number as three bf16 parts h + m + l (24 bits),
// products hh, hm, mh, mm, hl, lh per dimension (the dropped ml / lm / ll terms are below 2^-24 relative; bf16 x bf16 is exact in the fp32
// accumulator) = 24 slots — and |e_j|^2 as three parts against 1.0 = 3 slots.  The scalar kernel spends 4 fma + 3 compare / select per
// (pixel, code); here a lane receives 4 distances per MFMA, takes their minimum with v_min3 / v_min and enters the compare / select code
// only when some lane of the wave improved its best (a uniform branch that is rarely taken after the first code tiles).
// First minimum wins as in torch.argmin: strict <, codes visited in ascending order, ties between lane groups resolved by index.
constexpr int VQ_MAXC = 8192, VQ_PT = 8;            // codes the packed operand holds; pixel tiles (of 16) per wave
__device__ u32x4 g_vq_pack[VQ_MAXC / 16 * 64];     // A operand: [code tile][lane] 8 bf16 = K slots 8 kq .. 8 kq + 7 of code 16 tile + (lane & 15)

__device__ __forceinline__ void split3(float x, bf16& h, bf16& m, bf16& l) {
    h = (bf16)x;
    const float r = x - (float)h;
    m = (bf16)r;
    l = (bf16)(r - (float)m);
}

// K-slot layout (both operands): slot 4 t + d, t = 0..5, d = 0..3 — code side -2 x (eh, eh, em, em, eh, el)[d], pixel side (zh, zm, zh, zm, zl, zh)[d];
// slots 24, 25, 26: code side the three parts of |e|^2, pixel side 1; slots 27..31 zero
__global__ __launch_bounds__(256) void vq_pack_kernel(const float* cb, const float* esq, int ncodes, int ntiles) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntiles * 64) return;
    const int lane = i & 63, code = (i >> 6) * 16 + (lane & 15), kq = lane >> 4;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (code < ncodes) {
        bf16 h[4], m[4], l[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(-2.0f * cb[(size_t)code * 4 + d], h[d], m[d], l[d]);
        if (kq == 3) {
            bf16 a, b, c;
            split3(esq[code], a, b, c);
            v[0] = a; v[1] = b; v[2] = c;
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                v[d] = kq == 0 ? h[d] : (kq == 1 ? m[d] : h[d]);          // t = 0, 2, 4
                v[4 + d] = kq == 0 ? h[d] : (kq == 1 ? m[d] : l[d]);      // t = 1, 3, 5
            }
        }
    } else if (kq == 3) v[0] = (bf16)3.0e38f;                            // padding codes never win
    g_vq_pack[i] = __builtin_bit_cast(u32x4, v);
}

__global__ __launch_bounds__(256, 4) void vq_mfma_kernel(const float* z, const float* cb, int HW, int ntiles, float* q, int64_t* idx, size_t npix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const size_t p0 = ((size_t)blockIdx.x * 4 + wave) * (VQ_PT * 16);
    // pixel operands of this wave: VQ_PT tiles of 16 pixels
    bf16x8 zb[VQ_PT];
#pragma unroll
    for (int t = 0; t < VQ_PT; ++t) {
        const size_t i = p0 + t * 16 + n, ic = i < npix ? i : npix - 1;
        const size_t b = ic / HW, pix = ic % HW;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (kq == 3) {
            v[0] = (bf16)1.0f; v[1] = (bf16)1.0f; v[2] = (bf16)1.0f;
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16 h, m, l;
                split3(z[(b * 4 + d) * HW + pix], h, m, l);
                v[d] = kq == 2 ? l : h;                                   // t = 0 (zh), 2 (zh), 4 (zl)
                v[4 + d] = kq == 2 ? h : m;                               // t = 1 (zm), 3 (zm), 5 (zh)
            }
        }
        zb[t] = v;
    }
    float best[VQ_PT];
    int bi[VQ_PT];
#pragma unroll
    for (int t = 0; t < VQ_PT; ++t) { best[t] = INFINITY; bi[t] = 0; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const u32x4* pk = g_vq_pack + lane;
    // r04: software pipeline over HALF code tiles (4 of the wave's 8 pixel tiles): the matrix pipe works on one half while the other half's
    // distances are examined, and a half costs ONE branch — per pixel tile min / min / min / compare, the compare masks OR-ed on the scalar
    // unit; only when some lane of the wave improved do the flagged pixel tiles (scalar branches on masks that already exist) run their
    // compare / select code.  The first form (MFMA -> min -> compare -> branch per pixel tile: 8 dependent round trips per code tile) ran at
    // 68 cycles per MFMA: 535 us at 524 k pixels x 8192 codes.
    constexpr int HT = VQ_PT / 2;
    auto examine = [&](const f32x4* d, int ct, int t0) {
        const int c0 = ct * 16 + 4 * kq;                                  // this lane's four codes of the tile
        unsigned long long fl[HT], any = 0ull;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            // (fminf costs a v_max x, x per operand here — sNaN quieting; the distances are finite.  These reads stand a whole examine() and
            // four more MFMAs behind the MFMAs that wrote d: far beyond the 18 wait states a VALU read of an XDL result needs)
            float m3, m4;
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(d[t][0]), "v"(d[t][1]), "v"(d[t][2]));
            asm("v_min_f32 %0, %1, %2" : "=v"(m4) : "v"(m3), "v"(d[t][3]));
            fl[t] = __ballot(m4 < best[t0 + t]);
            any |= fl[t];
        }
        if (any) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                if (fl[t]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool lt = d[t][r] < best[t0 + t];
                        best[t0 + t] = lt ? d[t][r] : best[t0 + t];
                        bi[t0 + t] = lt ? c0 + r : bi[t0 + t];
                    }
                }
            }
        }
    };
    f32x4 dA[HT], dB[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) dB[t] = f32x4{INFINITY, INFINITY, INFINITY, INFINITY};      // (the first examine of the B half has no tile yet)
    // code tiles four ahead in registers (an iteration is ~250 cycles, an L2 round trip 500 - 800: two ahead stalled every iteration); past the
    // last tile the last one is examined again — strict < never takes a distance twice
    u32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = pk[(size_t)(j < ntiles ? j : ntiles - 1) * 64];
    for (int ct0 = 0; ct0 < ntiles; ct0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ct = ct0 + j < ntiles ? ct0 + j : ntiles - 1;
            const bf16x8 A = __builtin_bit_cast(bf16x8, a[j]);
            a[j] = pk[(size_t)(ct0 + j + 4 < ntiles ? ct0 + j + 4 : ntiles - 1) * 64];
#pragma unroll
            for (int t = 0; t < HT; ++t) dA[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, zb[t], zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            examine(dB, ct0 + j - 1 < ntiles ? ct0 + j - 1 : ntiles - 1, HT);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < HT; ++t) dB[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, zb[HT + t], zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            examine(dA, ct, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    examine(dB, ntiles - 1, HT);
    // the four lane groups of a pixel hold disjoint code subsets: smaller distance wins, equal distances -> smaller index
#pragma unroll
    for (int t = 0; t < VQ_PT; ++t) {
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float ob = __shfl_xor(best[t], sh, 64);
            const int oi = __shfl_xor(bi[t], sh, 64);
            const bool take = ob < best[t] || (ob == best[t] && oi < bi[t]);
            best[t] = take ? ob : best[t];
            bi[t] = take ? oi : bi[t];
        }
        const size_t i = p0 + t * 16 + n;
        if (kq == 0 && i < npix) {
            const size_t b = i / HW, pix = i % HW;
            if (idx) idx[i] = bi[t];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float zv = z[(b * 4 + c) * HW + pix], e = cb[(size_t)bi[t] * 4 + c];
                q[(b * 4 + c) * HW + pix] = zv + (e - zv);                // straight-through form of VQGAN.py:140
            }
        }
    }
}

template <typename T>
__global__ void decoder_tail_kernel(const T* x, int Cs, int HW, float* out, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, pix = i % HW;
        const T* px = x + i * Cs;
        const float v0 = to_f32(px[0]), v1 = to_f32(px[1]), v2 = to_f32(px[2]);
        out[(b * 3 + 0) * HW + pix] = v0 > 20.f ? v0 : log1pf(expf(v0));  // F.softplus (beta 1, threshold 20)
        out[(b * 3 + 1) * HW + pix] = tanhf(v1);
        out[(b * 3 + 2) * HW + pix] = tanhf(v2);
    }
}

// ------------------------------------------------------------------------------------------------ iSTFT
// One block per (frame, sample): rebuild the Hermitian spectrum from ISTFT+ (mag = expm1(c0),
// phase = atan2(sin, cos); the DC row is zero: tools.py:185-191), 1024-point radix-2 inverse FFT in LDS,
// multiply by the periodic Hann window, store the frame.
constexpr int NFFT = 1024;
constexpr int IF_FR = 4;      // frames per block
// One block = IF_FR consecutive frames of one sample.  The representation is [3][F][T] (time fastest): a block that owns
// ONE frame reads its 3 x 512 values with a stride of T floats (a 64-byte line per 4 useful bytes).  With four frames per
// block a thread fetches one 16-byte piece (4 frames of one bin) per channel, the twiddle factors e^{2 pi i j / 1024} are
// tabulated once per block in LDS instead of one sincospif per butterfly, and the four FFTs share every index computation.
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* enc, int F, int T, float* frames) {
    __shared__ float re[IF_FR][NFFT], im[IF_FR][NFFT];
    __shared__ float twc[NFFT / 2], tws[NFFT / 2];
    const int t0 = blockIdx.x * IF_FR, b = blockIdx.y, tid = threadIdx.x;
    const float* e0 = enc + (size_t)b * 3 * F * T;
    for (int j = tid; j < NFFT / 2; j += 256) sincospif(2.0f * (float)j / (float)NFFT, &tws[j], &twc[j]);
    // (every offset below is a multiple of 4 floats when T % 4 == 0, so the 16-byte loads are aligned iff enc is: a view with an odd
    // storage offset takes the scalar path)
    const bool vec_ok = (T % 4 == 0) && t0 + IF_FR <= T && (reinterpret_cast<uintptr_t>(enc) & 15) == 0;
    // bins 1..F from rows 0..F-1; bin 0 = 0; bins F+1..2F-1 by conjugate symmetry.  Stored bit-reversed.
    for (int kk = tid + 1; kk <= F; kk += 256) {
        float c0[IF_FR], c1[IF_FR], c2[IF_FR];
        const size_t o = (size_t)(kk - 1) * T + t0;
        if (vec_ok) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(e0 + o), c = *reinterpret_cast<const f32x4*>(e0 + (size_t)F * T + o),
                        s = *reinterpret_cast<const f32x4*>(e0 + 2 * (size_t)F * T + o);
#pragma unroll
            for (int f = 0; f < IF_FR; ++f) { c0[f] = a[f]; c1[f] = c[f]; c2[f] = s[f]; }
        } else {
#pragma unroll
            for (int f = 0; f < IF_FR; ++f) {
                const bool ok = t0 + f < T;
                c0[f] = ok ? e0[o + f] : 0.f;
                c1[f] = ok ? e0[(size_t)F * T + o + f] : 1.f;
                c2[f] = ok ? e0[2 * (size_t)F * T + o + f] : 0.f;
            }
        }
        const int r = __brev((unsigned)kk) >> 22, rm = __brev((unsigned)(NFFT - kk)) >> 22;   // 10-bit reversal
#pragma unroll
        for (int f = 0; f < IF_FR; ++f) {
            // mag * exp(i atan2(sin, cos)) = mag * (cos, sin) / |(cos, sin)| — no atan2f / cosf / sinf (three libm calls per bin were half of
            // this kernel's 168 us); atan2(0, 0) = 0 in the reference's formula: (mag, 0)
            const float mag = expm1f(c0[f]);
            // (the pair is first scaled by an exact power of two so that max(|cos|, |sin|) lies in [0.5, 1): c^2 + s^2 neither underflows —
            // v_rsq_f32 returns +inf for a denormal argument, and a pair of 1e-23s used to lose its phase to n2 == 0 — nor overflows)
            const float pm = fmaxf(fabsf(c1[f]), fabsf(c2[f]));
            const bool z0 = !(pm > 0.f) || !(pm < 3.0e38f);               // 0 (and nothing finite): the reference's phase is 0 (resp. undefined)
            const int pe = z0 ? 0 : __builtin_amdgcn_frexp_expf(pm);
            const float pa = ldexpf(c1[f], -pe), pb = ldexpf(c2[f], -pe);
            const float n2 = pa * pa + pb * pb;                           // in [0.25, 2)
            float ri = __builtin_amdgcn_rsqf(n2);
            ri = ri * (1.5f - 0.5f * n2 * ri * ri);                       // one Newton step: v_rsq_f32 alone is ~1 ulp
            const float xr = z0 ? mag : mag * (pa * ri), xi = z0 ? 0.f : mag * (pb * ri);
            re[f][r] = xr;
            im[f][r] = xi;
            if (kk < F) {                    // the mirrored bin NFFT - kk (kk = F is its own mirror)
                re[f][rm] = xr;
                im[f][rm] = -xi;
            }
        }
    }
    if (tid < IF_FR) {
        re[tid][0] = 0.f;                    // DC (bit reversal of 0)
        im[tid][0] = 0.f;
    }
    __syncthreads();
    for (int len = 2, shift = 9; len <= NFFT; len <<= 1, --shift) {
        const int half = len >> 1;
        for (int j = tid; j < NFFT / 2; j += 256) {
            const int grp = j / half, pos = j - grp * half;
            const int i0 = grp * len + pos, i1 = i0 + half;
            const float c = twc[pos << shift], s = tws[pos << shift];      // e^{+2 pi i pos/len}: inverse transform
#pragma unroll
            for (int f = 0; f < IF_FR; ++f) {
                const float tr = re[f][i1] * c - im[f][i1] * s, ti = re[f][i1] * s + im[f][i1] * c;
                const float ur = re[f][i0], ui = im[f][i0];
                re[f][i0] = ur + tr; im[f][i0] = ui + ti;
                re[f][i1] = ur - tr; im[f][i1] = ui - ti;
            }
        }
        __syncthreads();
    }
    for (int n = tid; n < NFFT; n += 256) {
        const float w = 0.5f - 0.5f * (n < NFFT / 2 ? twc[n] : -twc[n - NFFT / 2]);   // periodic Hann from the same table
#pragma unroll
        for (int f = 0; f < IF_FR; ++f)
            if (t0 + f < T) frames[((size_t)b * T + t0 + f) * NFFT + n] = re[f][n] * (1.0f / NFFT) * w;
    }
}

// ---- r04: the same frames by a radix-4 Stockham transform.  The radix-2 kernel above is bound by its LDS traffic: ten passes of 4-byte
// reads / writes, 200 LDS operations per thread and frame, 164 us at 64 x 256 frames.  Here a thread owns ONE radix-4 butterfly per frame and
// pass: five passes between two buffers (autosort: natural order in, natural order out, no bit reversal), 8-byte complex elements, reads
// contiguous across the wave, three twiddles from one table entry (w, w^2, w^3) — 40 LDS operations per thread and frame.
constexpr int R4_LDS = 2 * IF_FR * NFFT * 8 + 256 * 8;
__global__ __launch_bounds__(256) void istft_frames_r4_kernel(const float* enc, int F, int T, float* frames) {
    extern __shared__ __attribute__((aligned(16))) float2 r4sm[];
    float2* const buf0 = r4sm;                                        // [IF_FR][NFFT]
    float2* const buf1 = r4sm + IF_FR * NFFT;
    float2* const tw = r4sm + 2 * IF_FR * NFFT;                        // exp(2 pi i m / 1024), m < 256
    const int t0 = blockIdx.x * IF_FR, b = blockIdx.y, tid = threadIdx.x;
    const float* e0 = enc + (size_t)b * 3 * F * T;
    {
        float sn, cs;
        sincospif(2.0f * (float)tid / (float)NFFT, &sn, &cs);
        tw[tid] = float2{cs, sn};
    }
    const bool vec_ok = (T % 4 == 0) && t0 + IF_FR <= T && (reinterpret_cast<uintptr_t>(enc) & 15) == 0;
    // bins 1..F from rows 0..F-1; bin 0 = 0; bins F+1..2F-1 by conjugate symmetry (tools.py:185-191)
    for (int kk = tid + 1; kk <= F; kk += 256) {
        float c0[IF_FR], c1[IF_FR], c2[IF_FR];
        const size_t o = (size_t)(kk - 1) * T + t0;
        if (vec_ok) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(e0 + o), c = *reinterpret_cast<const f32x4*>(e0 + (size_t)F * T + o),
                        sv = *reinterpret_cast<const f32x4*>(e0 + 2 * (size_t)F * T + o);
#pragma unroll
            for (int f = 0; f < IF_FR; ++f) { c0[f] = a[f]; c1[f] = c[f]; c2[f] = sv[f]; }
        } else {
#pragma unroll
            for (int f = 0; f < IF_FR; ++f) {
                const bool ok = t0 + f < T;
                c0[f] = ok ? e0[o + f] : 0.f;
                c1[f] = ok ? e0[(size_t)F * T + o + f] : 1.f;
                c2[f] = ok ? e0[2 * (size_t)F * T + o + f] : 0.f;
            }
        }
#pragma unroll
        for (int f = 0; f < IF_FR; ++f) {
            const float mag = expm1f(c0[f]);
            const float pm = fmaxf(fabsf(c1[f]), fabsf(c2[f]));           // (power-of-two prescale: see the radix-2 kernel above)
            const bool z0 = !(pm > 0.f) || !(pm < 3.0e38f);
            const int pe = z0 ? 0 : __builtin_amdgcn_frexp_expf(pm);
            const float pa = ldexpf(c1[f], -pe), pb = ldexpf(c2[f], -pe);
            const float n2 = pa * pa + pb * pb;
            float ri = __builtin_amdgcn_rsqf(n2);
            ri = ri * (1.5f - 0.5f * n2 * ri * ri);
            const float xr = z0 ? mag : mag * (pa * ri), xi = z0 ? 0.f : mag * (pb * ri);
            buf0[f * NFFT + kk] = float2{xr, xi};
            if (kk < F) buf0[f * NFFT + NFFT - kk] = float2{xr, -xi};
        }
    }
    if (tid < IF_FR) buf0[tid * NFFT] = float2{0.f, 0.f};            // DC
    __syncthreads();
    auto cmul = [](float2 a, float2 w) { return float2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; };
    auto pass = [&](const float2* x, float2* y, int p, int sh) {       // p = 4^s, sh = log2(256 / p)
        const int k = tid & (p - 1), j = ((tid - k) << 2) + k;
        const float2 w1 = tw[k << sh], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
        for (int f = 0; f < IF_FR; ++f) {
            const float2* xf = x + f * NFFT;
            const float2 u0 = xf[tid], u1 = cmul(xf[tid + 256], w1), u2 = cmul(xf[tid + 512], w2), u3 = cmul(xf[tid + 768], w3);
            const float2 a0 = {u0.x + u2.x, u0.y + u2.y}, a1 = {u0.x - u2.x, u0.y - u2.y}, a2 = {u1.x + u3.x, u1.y + u3.y};
            const float2 a3 = {-(u1.y - u3.y), u1.x - u3.x};          // i (u1 - u3): the inverse transform's sign
            float2* yf = y + f * NFFT + j;
            yf[0] = float2{a0.x + a2.x, a0.y + a2.y};
            yf[p] = float2{a1.x + a3.x, a1.y + a3.y};
            yf[2 * p] = float2{a0.x - a2.x, a0.y - a2.y};
            yf[3 * p] = float2{a1.x - a3.x, a1.y - a3.y};
        }
        __syncthreads();
    };
    pass(buf0, buf1, 1, 8);
    pass(buf1, buf0, 4, 6);
    pass(buf0, buf1, 16, 4);
    pass(buf1, buf0, 64, 2);
    pass(buf0, buf1, 256, 0);
    for (int n = tid; n < NFFT; n += 256) {
        const int m = n & 255, qd = n >> 8;
        const float2 e = tw[m];
        const float cs = qd == 0 ? e.x : (qd == 1 ? -e.y : (qd == 2 ? -e.x : e.y));     // cos(2 pi n / 1024)
        const float w = 0.5f - 0.5f * cs;                                              // periodic Hann
#pragma unroll
        for (int f = 0; f < IF_FR; ++f)
            if (t0 + f < T) frames[((size_t)b * T + t0 + f) * NFFT + n] = buf1[f * NFFT + n].x * (1.0f / NFFT) * w;
    }
}

__global__ void istft_ola_kernel(const float* frames, int T, int hop, float* audio, int L) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (s >= L) return;
    const int pos = s + NFFT / 2;
    int t_hi = pos / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    int t_lo = (pos - NFFT + hop) / hop;  // smallest t with t*hop > pos - NFFT
    if (pos - NFFT < 0) t_lo = 0;
    float y = 0.f, wss = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int n = pos - t * hop;
        if (n < 0 || n >= NFFT) continue;
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)NFFT);
        y += frames[((size_t)b * T + t) * NFFT + n];
        wss += w * w;
    }
    audio[(size_t)b * L + s] = wss > 1.17549435e-38f ? y / wss : y;
}

// ------------------------------------------------------------------------------------------------ STFT+
// One block per (frame, sample): gather the centred 1024-sample frame (zero or reflect padding), periodic Hann,
// forward radix-2 FFT in LDS, then bins 1..512 -> (log1p|X|, cos arg X, sin arg X).  Columns past the signal's
// frames are the encoding of a zero spectrum (0, 1, 0), i.e. tools.pad_STFT followed by tools.encode_stft.
__global__ __launch_bounds__(256) void stft_plus_kernel(const float* audio, int L, int hop, int reflect, int T, int T_out, float* enc) {
    __shared__ float re[NFFT], im[NFFT];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int F = NFFT / 2;
    float* e0 = enc + (size_t)b * 3 * F * T_out;
    if (t >= T) {
        for (int k = tid; k < F; k += 256) {
            e0[(size_t)k * T_out + t] = 0.f;
            e0[((size_t)F + k) * T_out + t] = 1.f;
            e0[((size_t)2 * F + k) * T_out + t] = 0.f;
        }
        return;
    }
    const float* y = audio + (size_t)b * L;
    for (int n = tid; n < NFFT; n += 256) {
        int i = t * hop + n - NFFT / 2;
        float v = 0.f;
        if (reflect) {
            if (i < 0) i = -i;
            if (i >= L) i = 2 * (L - 1) - i;
            if (i >= 0 && i < L) v = y[i];
        } else if (i >= 0 && i < L) {
            v = y[i];
        }
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)NFFT);
        const int r = __brev((unsigned)n) >> 22;
        re[r] = v * w;
        im[r] = 0.f;
    }
    __syncthreads();
    for (int len = 2; len <= NFFT; len <<= 1) {
        const int half = len >> 1;
        for (int j = tid; j < NFFT / 2; j += 256) {
            const int grp = j / half, pos = j % half;
            const int i0 = grp * len + pos, i1 = i0 + half;
            float s, c;
            sincospif(-2.0f * (float)pos / (float)len, &s, &c);  // e^{-2 pi i pos/len}: forward transform
            const float tr = re[i1] * c - im[i1] * s, ti = re[i1] * s + im[i1] * c;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    for (int k = tid; k < F; k += 256) {
        const float xr = re[k + 1], xi = im[k + 1];
        const float mag = sqrtf(xr * xr + xi * xi);
        e0[(size_t)k * T_out + t] = log1pf(mag);
        e0[((size_t)F + k) * T_out + t] = mag > 0.f ? xr / mag : 1.f;
        e0[((size_t)2 * F + k) * T_out + t] = mag > 0.f ? xi / mag : 0.f;
    }
}

}  // namespace

extern "C" int ds_stft_plus(const float* audio, int B, int L, int hop, int reflect_pad, int T_out, float* enc, void* stream) {
    DS_REQUIRE(audio && enc && B > 0 && L > 0 && hop > 0, "stft_plus: bad args");
    const int T = 1 + L / hop;
    DS_REQUIRE(T_out >= T, "stft_plus: T_out=%d is smaller than the %d frames of the signal", T_out, T);
    DS_REQUIRE(!reflect_pad || L > NFFT / 2, "stft_plus: reflect padding needs more than %d samples", NFFT / 2);
    hipLaunchKernelGGL(stft_plus_kernel, dim3(T_out, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), audio, L, hop, reflect_pad,
                       T, T_out, enc);
    DS_CHECK_LAUNCH("stft_plus");
    return DS_OK;
}

extern "C" int ds_vq_nearest(const float* z, const float* cb, const float* esq, int B, int D, int HW, int ncodes, float* q,
                             int64_t* idx, void* stream) {
    DS_REQUIRE(z && cb && esq && q && B > 0 && HW > 0 && ncodes > 0, "vq_nearest: bad args");
    DS_REQUIRE(D == 4, "vq_nearest: embedding_dim %d unsupported (4 only)", D);
    const size_t npix = (size_t)B * HW;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool scalar = getenv("DS_VQ_SCALAR") != nullptr;          // A/B switch: the r01 kernel
    if (ncodes <= VQ_MAXC && !scalar) {
        // matrix-core search: the codebook is re-packed per call (it is a module parameter: no caching across calls), 3 us
        const int ntiles = (ncodes + 15) / 16;
        hipLaunchKernelGGL(vq_pack_kernel, dim3((ntiles * 64 + 255) / 256), dim3(256), 0, st, cb, esq, ncodes, ntiles);
        DS_CHECK_LAUNCH("vq_pack");
        const size_t per_block = 4 * VQ_PT * 16;
        hipLaunchKernelGGL(vq_mfma_kernel, dim3((unsigned)((npix + per_block - 1) / per_block)), dim3(256), 0, st, z, cb, HW, ntiles, q, idx, npix);
        DS_CHECK_LAUNCH("vq_nearest");
        return DS_OK;
    }
    hipLaunchKernelGGL(vq_kernel<4>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, z, cb, esq, HW, ncodes, q, idx, npix);
    DS_CHECK_LAUNCH("vq_nearest");
    return DS_OK;
}

// ---- the quantiser's two scalars (VQGAN.py:62-73 / :131-144): mse = mean((q - z)^2) and perplexity = exp(-sum p log(p + 1e-10)), p = code
// usage frequencies — as torch ops this was a dozen launches and a host synchronisation inside torch.bincount per forward
namespace {
constexpr int VQS_BLOCKS = 2048;
__global__ __launch_bounds__(256) void vq_stats_kernel(const float* z, const float* q, const int64_t* idx, int D, int HW, size_t npix, int ncodes,
                                                        unsigned* counts, double* part) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW, pix = i % HW;
        float a = 0.f;
        for (int c = 0; c < D; ++c) {
            const float d = q[(b * D + c) * HW + pix] - z[(b * D + c) * HW + pix];
            a = fmaf(d, d, a);
        }
        s += (double)a;
        const int64_t j = idx[i];
        if (j >= 0 && j < ncodes) atomicAdd(counts + j, 1u);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(1024) void vq_stats_finish_kernel(const unsigned* counts, const double* part, int nparts, int ncodes, double n_elems,
                                                               double n_pix, float cc, int ema, float* out2) {
    __shared__ double red[16];
    double h = 0.0;
    for (int j = threadIdx.x; j < ncodes; j += 1024) {
        const float pr = (float)((double)counts[j] / n_pix);
        h += (double)(pr * logf(pr + 1e-10f));
    }
    double s = 0.0;
    for (int j = threadIdx.x; j < nparts; j += 1024) s += part[j];
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        h += __shfl_xor(h, sh, 64);
        s += __shfl_xor(s, sh, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = h;
    __syncthreads();
    double ht = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < 16; ++w) ht += red[w];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double st = 0.0;
        for (int w = 0; w < 16; ++w) st += red[w];
        const float mse = (float)(st / n_elems);
        out2[0] = mse;
        out2[1] = expf((float)(-ht));
        // the module's loss (VQGAN.py:64-66 / :133-134): commitment_cost * e_latent_loss (EMA) or q_latent_loss + commitment_cost * e_latent_loss
        out2[2] = ema ? __fmul_rn(cc, mse) : __fadd_rn(mse, __fmul_rn(cc, mse));
    }
}
}  // namespace

extern "C" size_t ds_vq_stats_ws_bytes(int ncodes) { return (size_t)ncodes * 4 + VQS_BLOCKS * 8 + 8; }

extern "C" int ds_vq_stats(const float* z, const float* q, const int64_t* idx, int B, int D, int HW, int ncodes, float commitment_cost, int ema,
                           float* out2, void* ws, void* stream) {
    DS_REQUIRE(z && q && idx && out2 && ws && B > 0 && D > 0 && HW > 0 && ncodes > 0, "vq_stats: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t npix = (size_t)B * HW;
    double* part = reinterpret_cast<double*>(ws);                               // [VQS_BLOCKS] then the counts
    unsigned* counts = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + VQS_BLOCKS * 8);
    if (hipMemsetAsync(counts, 0, (size_t)ncodes * 4, st) != hipSuccess) DS_FAIL(DS_ELAUNCH, "vq_stats: memset failed");
    const int blocks = (int)((npix + 255) / 256 < VQS_BLOCKS ? (npix + 255) / 256 : VQS_BLOCKS);
    hipLaunchKernelGGL(vq_stats_kernel, dim3(blocks), dim3(256), 0, st, z, q, idx, D, HW, npix, ncodes, counts, part);
    DS_CHECK_LAUNCH("vq_stats");
    hipLaunchKernelGGL(vq_stats_finish_kernel, dim3(1), dim3(1024), 0, st, counts, part, blocks, ncodes, (double)npix * D, (double)npix, commitment_cost, ema, out2);
    DS_CHECK_LAUNCH("vq_stats_finish");
    return DS_OK;
}

extern "C" int ds_decoder_tail(const void* x, int dtype, int B, int Cs, int HW, float* out, void* stream) {
    DS_REQUIRE(x && out && B > 0 && Cs >= 3 && HW > 0, "decoder_tail: bad args");
    const size_t total = (size_t)B * HW;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DS_BF16) hipLaunchKernelGGL(decoder_tail_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, Cs, HW, out, total);
    else if (dtype == DS_F32) hipLaunchKernelGGL(decoder_tail_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, Cs, HW, out, total);
    else DS_FAIL(DS_EINVAL, "decoder_tail: dtype %d", dtype);
    DS_CHECK_LAUNCH("decoder_tail");
    return DS_OK;
}

extern "C" size_t ds_istft_ws_floats(int B, int F, int T) { return (size_t)B * T * 2 * F; }

extern "C" int ds_istft_plus(const float* enc, int B, int F, int T, int hop, float* ws, float* audio, void* stream) {
    DS_REQUIRE(enc && ws && audio && B > 0 && T > 1 && hop > 0, "istft_plus: bad args");
    DS_REQUIRE(2 * F == NFFT, "istft_plus: n_fft = 2*F must be %d (got F=%d)", NFFT, F);
    DS_REQUIRE(NFFT % hop == 0, "istft_plus: hop %d must divide n_fft", hop);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool r2 = getenv("DS_ISTFT_R2") != nullptr;              // A/B switch: the radix-2 kernel
    if (r2) hipLaunchKernelGGL(istft_frames_kernel, dim3((T + IF_FR - 1) / IF_FR, B), dim3(256), 0, st, enc, F, T, ws);
    else {
        DS_SET_MAX_LDS(istft_frames_r4_kernel, R4_LDS, "istft_frames_r4");
        hipLaunchKernelGGL(istft_frames_r4_kernel, dim3((T + IF_FR - 1) / IF_FR, B), dim3(256), R4_LDS, st, enc, F, T, ws);
    }
    DS_CHECK_LAUNCH("istft_frames");
    const int L = hop * (T - 1);
    hipLaunchKernelGGL(istft_ola_kernel, dim3((L + 255) / 256, B), dim3(256), 0, st, ws, T, hop, audio, L);
    DS_CHECK_LAUNCH("istft_ola");
    return DS_OK;
}
