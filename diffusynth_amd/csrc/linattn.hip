// Linear ("efficient") attention over a precomputed qkv tensor [B][N][3*heads*32] (gfx950).
//   ctx[d][e] = sum_n softmax_n(k)[d][n] v[e][n]      (32x32 per head; no N x N matrix ever exists)
//   out[e][n] = sum_d ctx[d][e] q~[d][n],  q~ = softmax_d(q + label_q) * scale   (or raw q)
// pass 1 splits N into segments (per-segment max / sum / unnormalised context), `combine` merges
// them with the usual max-rescaling (and appends linear_cat's extra key/value token), pass 2 streams
// q once.  HBM-bound: each qkv element is read exactly once per pass, fp32 math.
#include "common.hpp"

namespace {

constexpr int PART = 32 + 32 + 1024;
constexpr int TP = 64;  // pixels per LDS tile in pass 1

template <typename T>
__global__ __launch_bounds__(256) void attn_ctx_partial(const ds_attn_params p) {
    constexpr int V = Vec16<T>::N;
    constexpr int DV = 32 / V;  // vectors per 32 channels
    __shared__ float kt[TP][32];
    __shared__ __attribute__((aligned(16))) float vt[TP][32];
    __shared__ float smax[256 / DV][33];
    __shared__ float kmax[32];
    const int seg = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int HD = p.heads * 32, CQ = 3 * HD;
    const int per = (p.N + p.nseg - 1) / p.nseg;
    const int n0 = seg * per, n1 = min(p.N, n0 + per);
    const T* qkv = reinterpret_cast<const T*>(p.qkv) + (size_t)b * p.N * CQ;
    const int koff = HD + h * 32, voff = 2 * HD + h * 32;

    // sweep 1: per-d maximum of k over the segment
    {
        const int dv = tid % DV, pl = tid / DV;
        float mx[V];
#pragma unroll
        for (int v = 0; v < V; ++v) mx[v] = -INFINITY;
        // four rows per trip with every load in flight before the first is used (a rolled loop pays one memory round trip per row); rows past
        // the segment are clamped to its last row: a duplicate does not change a maximum
        for (int n = n0 + pl; n < n1; n += 4 * (256 / DV)) {
            float kv[4][V];
#pragma unroll
            for (int j = 0; j < 4; ++j) vec16_load<T>(qkv + (size_t)min(n + j * (256 / DV), n1 - 1) * CQ + koff + dv * V, kv[j], DS_BX_SRC0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < V; ++v) mx[v] = fmaxf(mx[v], kv[j][v]);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) smax[pl][dv * V + v] = mx[v];
        __syncthreads();
        if (tid < 32) {
            float m = -INFINITY;
            for (int r = 0; r < 256 / DV; ++r) m = fmaxf(m, smax[r][tid]);
            kmax[tid] = m;
        }
        __syncthreads();
    }

    // sweep 2: accumulate exp(k - max) and exp(k - max) v^T.  The 32 x 32 context of the tile is P^T V with P [64 pixels][32 d], V [64][32 e] in LDS:
    // on the fp32 matrix pipe (v_mfma_f32_32x32x2f32: exact fp32 products and sums) each wave takes 16 of the tile's pixels — 8 MFMAs against
    // 64 x (2 LDS reads + 5 VALU instructions) per thread of the VALU form, which bounded the kernel; the four partial contexts are added once
    // at the end.  The row sums l[d] are accumulated where the exponentials are computed (a thread's channels do not change from tile to tile).
    __shared__ __attribute__((aligned(16))) float cred[4][1024];
    __shared__ float lsum[32];
    const int lane = tid & 63, wave = tid >> 6;
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
    float lpart[V];
#pragma unroll
    for (int v = 0; v < V; ++v) lpart[v] = 0.f;
    static_assert(256 % (2 * DV) == 0, "a thread's (channel vector, k / v) role is the same in every staging iteration");
    for (int t0 = n0; t0 < n1; t0 += TP) {
        const int cnt = min(TP, n1 - t0);
        // the tile's pieces of this thread are all requested (rows past the segment clamped to its last row) before the first is converted
        constexpr int SIT = TP * 2 * DV / 256;
        float val[SIT][V];
#pragma unroll
        for (int s = 0; s < SIT; ++s) {
            const int i = tid + s * 256, dv = i % DV, which = (i / DV) & 1, pix = i / (2 * DV);
            vec16_load<T>(qkv + (size_t)(t0 + min(pix, cnt - 1)) * CQ + (which ? voff : koff) + dv * V, val[s], DS_BX_SRC0);
        }
#pragma unroll
        for (int s = 0; s < SIT; ++s) {
            const int i = tid + s * 256, dv = i % DV, which = (i / DV) & 1, pix = i / (2 * DV);
            const bool in = pix < cnt;
            if (which) {
#pragma unroll
                for (int v = 0; v < V; ++v) vt[pix][dv * V + v] = in ? val[s][v] : 0.f;
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float pv = in ? expf(val[s][v] - kmax[dv * V + v]) : 0.f;
                    kt[pix][dv * V + v] = pv;
                    lpart[v] += pv;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int n = 16 * wave + 2 * s + (lane >> 5);         // lane = (k = pixel n, row d / column e = lane & 31)
            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[n][lane & 31], vt[n][lane & 31], cacc, 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator register r of lane (fh = lane >> 5, column e = lane & 31) is row d = (r & 3) + 8 (r >> 2) + 4 fh
#pragma unroll
    for (int r = 0; r < 16; ++r) cred[wave][((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = cacc[r];
    // row sums: the 256 / (2 DV) threads that staged k pieces of a channel vector park their partial sums in the (free) sweep-1 buffer and one
    // thread per channel adds them in a fixed order (no float atomics: the result must not depend on arrival order)
    if (((tid / DV) & 1) == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) smax[tid / (2 * DV)][(tid % DV) * V + v] = lpart[v];
    }
    __syncthreads();
    if (tid < 32) {
        float a = 0.f;
        for (int g = 0; g < 256 / (2 * DV); ++g) a += smax[g][tid];
        lsum[tid] = a;
    }
    __syncthreads();
    float* out = p.part + (((size_t)b * p.heads + h) * p.nseg + seg) * PART;
    if (tid < 32) {
        out[tid] = kmax[tid];
        out[32 + tid] = lsum[tid];
    }
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(&cred[0][4 * tid]), c1 = *reinterpret_cast<const f32x4*>(&cred[1][4 * tid]),
                c2 = *reinterpret_cast<const f32x4*>(&cred[2][4 * tid]), c3 = *reinterpret_cast<const f32x4*>(&cred[3][4 * tid]);
    *reinterpret_cast<f32x4*>(out + 64 + 4 * tid) = f32x4{(c0[0] + c1[0]) + (c2[0] + c3[0]), (c0[1] + c1[1]) + (c2[1] + c3[1]),
                                                          (c0[2] + c1[2]) + (c2[2] + c3[2]), (c0[3] + c1[3]) + (c2[3] + c3[3])};
}

// Thread = element (d, e) of ctx.  r04: the per-(segment, d) factors exp(m_s - M) and the row sums are computed ONCE per row — thread (d, e)
// takes segment s0 + e of a batch of 32 — and shared through LDS; before, each of the 32 threads of a row evaluated every segment's expf itself
// (128 segments at a small batch: 15.8 us per launch, 16 launches per forward).  The sums run over the segments in ascending order as before.
// Four blocks per (sample, head), eight rows each: one block pulled its 128 x 4 KB of partial contexts through ONE CU's L2 port (12 us at batch 1).
__global__ __launch_bounds__(256) void attn_ctx_combine(const ds_attn_params p) {
    __shared__ float sf[32][9], sl[32][9];                      // [segment of the batch][row of this block]
    const int h = blockIdx.x, b = blockIdx.y, dl = threadIdx.x >> 5, d = blockIdx.z * 8 + dl, e = threadIdx.x & 31;
    const float* part = p.part + ((size_t)b * p.heads + h) * p.nseg * PART;
    const bool tok = p.label_k != nullptr;
    float lk = 0.f, lv = 0.f;
    if (tok) {
        lk = p.label_k[(size_t)b * p.lk_stride + h * 32 + d];
        lv = p.label_v[(size_t)b * p.lv_stride + h * 32 + e];
    }
    // row maximum: lane e looks at segments e, e + 32, ..; the 32 lanes of a row are one half of a wave
    float M = tok ? lk : -INFINITY;
    for (int s = e; s < p.nseg; s += 32) M = fmaxf(M, part[(size_t)s * PART + d]);
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) M = fmaxf(M, __shfl_xor(M, sh, 64));
    float Lr = tok ? expf(lk - M) : 0.f;
    float A = tok ? expf(lk - M) * lv : 0.f;
    for (int s0 = 0; s0 < p.nseg; s0 += 32) {
        const int nb = min(32, p.nseg - s0);
        float cv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) cv[j] = part[(size_t)min(s0 + j, p.nseg - 1) * PART + 64 + d * 32 + e];
        {
            const float* ps = part + (size_t)min(s0 + e, p.nseg - 1) * PART;
            sf[e][dl] = e < nb ? expf(ps[d] - M) : 0.f;
            sl[e][dl] = ps[32 + d];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float f = sf[j][dl];
            Lr += f * sl[j][dl];
            A += f * cv[j];
        }
        __syncthreads();
    }
    p.ctx[(((size_t)b * p.heads + h) * 32 + d) * 32 + e] = A / Lr;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_out_kernel(const ds_attn_params p) {
    constexpr int V = Vec16<T>::N;
    constexpr int DV = 32 / V;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // ctx[heads][32][32] | lq[heads*32]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int HD = p.heads * 32, CQ = 3 * HD;
    float* ctx = sm;
    float* lq = sm + p.heads * 1024;
    for (int i = tid; i < p.heads * 1024; i += 256) ctx[i] = p.ctx[(size_t)b * p.heads * 1024 + i];
    for (int i = tid; i < HD; i += 256) lq[i] = p.label_q ? p.label_q[(size_t)b * p.lq_stride + i] : 0.f;
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    // fp32: a thread's pixel row is 128 bytes per head, and with thread = pixel every load / store instruction is 64 separate 16-byte pieces
    // (1536 / 512 bytes apart): the L2's request rate bounds the kernel, not bytes.  Rows are therefore moved by 8 lanes per pixel (whole 128-byte
    // lines per instruction) and transposed through a wave-private LDS tile [64 pixels][36 floats] (pitch 144 B: conflict-free 16-byte rows).
    constexpr bool ROWS = sizeof(T) == 4 && !DS_BOUNDS;
    float* const stage = lq + HD + (tid >> 6) * (64 * 36);
    const int lane = tid & 63, nb = blockIdx.x * 256 + (tid & ~63);
    if (!ROWS && n >= p.N) return;
    const T* qrow = reinterpret_cast<const T*>(p.qkv) + ((size_t)b * p.N + n) * CQ;
    T* orow = reinterpret_cast<T*>(p.out) + ((size_t)b * p.N + n) * HD;
    for (int h = 0; h < p.heads; ++h) {
        float q[32];
        if constexpr (ROWS) {
            f32x4 pc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {                                // piece lane + 64 t of the wave's 64 rows: row (piece >> 3), 16-byte piece (piece & 7)
                const int id = lane + 64 * t, row = min(nb + (id >> 3), p.N - 1);
                pc[t] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.qkv) + ((size_t)b * p.N + row) * CQ + h * 32 + (id & 7) * 4);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int id = lane + 64 * t;
                *reinterpret_cast<f32x4*>(stage + (id >> 3) * 36 + (id & 7) * 4) = pc[t];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + lane * 36 + 4 * j);
                q[4 * j] = v[0]; q[4 * j + 1] = v[1]; q[4 * j + 2] = v[2]; q[4 * j + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int dv = 0; dv < DV; ++dv) vec16_load<T>(qrow + h * 32 + dv * V, q + dv * V, DS_BX_SRC0);
        }
        if (p.q_softmax) {
            float mx = -INFINITY;
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                q[d] += lq[h * 32 + d];
                mx = fmaxf(mx, q[d]);
            }
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                q[d] = expf(q[d] - mx);
                s += q[d];
            }
            const float inv = 1.0f / s;
#pragma unroll
            for (int d = 0; d < 32; ++d) q[d] = q[d] * inv * p.scale;
        }
        float o[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) o[e] = 0.f;
        const float* c = ctx + h * 1024;
        if constexpr (ROWS) {
            // the context is the same for every lane: read through the constant address space it arrives in SGPRs (scalar loads, one per
            // 16 values) and the 1024 fmas of a head take it as their scalar operand — as 256 broadcast ds_read_b128 per head the LDS pipe,
            // not memory, bounded the kernel (nothing writes p.ctx during this launch)
            typedef const float __attribute__((address_space(4))) * cptr_t;
            const unsigned long long ca = (unsigned long long)(p.ctx + ((size_t)b * p.heads + h) * 1024);
            const cptr_t cg = (cptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ca >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)ca));
#pragma unroll
            for (int d = 0; d < 32; ++d)
#pragma unroll
                for (int e = 0; e < 32; ++e) o[e] = fmaf(cg[d * 32 + e], q[d], o[e]);
        } else
#pragma unroll
        for (int d = 0; d < 32; ++d) {
#pragma unroll
            for (int e4 = 0; e4 < 8; ++e4) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(c + d * 32 + e4 * 4);
                o[e4 * 4 + 0] = fmaf(cv[0], q[d], o[e4 * 4 + 0]);
                o[e4 * 4 + 1] = fmaf(cv[1], q[d], o[e4 * 4 + 1]);
                o[e4 * 4 + 2] = fmaf(cv[2], q[d], o[e4 * 4 + 2]);
                o[e4 * 4 + 3] = fmaf(cv[3], q[d], o[e4 * 4 + 3]);
            }
        }
        if constexpr (ROWS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(stage + lane * 36 + 4 * j) = f32x4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int id = lane + 64 * t, row = nb + (id >> 3);
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (id >> 3) * 36 + (id & 7) * 4);
                if (row < p.N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((size_t)b * p.N + row) * HD + h * 32 + (id & 7) * 4) = v;
            }
        } else {
#pragma unroll
            for (int dv = 0; dv < DV; ++dv) vec16_store<T>(orow + h * 32 + dv * V, o + dv * V, DS_BX_OUT);
        }
    }
}

int check(const ds_attn_params* p) {
    DS_REQUIRE(p && p->qkv && p->part && p->ctx, "linattn: null pointer");
    DS_REQUIRE(p->dtype == DS_F32 || p->dtype == DS_BF16, "linattn: dtype %d", p->dtype);
    DS_REQUIRE(p->B > 0 && p->N > 0 && p->heads > 0 && p->heads <= 8 && p->nseg > 0 && p->nseg <= p->N, "linattn: bad sizes");
    DS_REQUIRE((p->label_k == nullptr) == (p->label_v == nullptr), "linattn: label_k and label_v go together");
    if (!ds_aligned16(p->qkv) || !ds_aligned16(p->part)) DS_FAIL(DS_EALIGN, "linattn: pointers must be 16-byte aligned");
    return DS_OK;
}

}  // namespace

#if DS_BOUNDS
static void linattn_publish_bounds(const ds_attn_params* p, hipStream_t st) {
    const long long es = p->dtype == DS_BF16 ? 2 : 4;
    DsBxHost h(DS_K_LINATTN);
    h.set(DS_BX_SRC0, p->qkv, (long long)p->B * p->N * 3 * p->heads * 32 * es);
    h.set(DS_BX_OUT, p->out, (long long)p->B * p->N * p->heads * 32 * es);
    h.publish(st);
}
extern "C" int ds_bounds_fetch_linattn(ds_bounds_rec* out, int reset) { return ds_bounds_fetch_tu(out, reset); }
#endif

int ds_linattn_launch_combine(const ds_attn_params* p, hipStream_t st) {   // shared with attn_fused.hip
#if DS_BOUNDS
    DsBxHost(0).publish(st);
#endif
    hipLaunchKernelGGL(attn_ctx_combine, dim3(p->heads, p->B, 4), dim3(256), 0, st, *p);
    DS_CHECK_LAUNCH("attn_ctx_combine");
    return DS_OK;
}

extern "C" size_t ds_linattn_part_floats(int B, int heads, int nseg) { return (size_t)B * heads * nseg * PART; }

extern "C" int ds_linattn_context(const ds_attn_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(p->nseg, p->heads, p->B);
#if DS_BOUNDS
    linattn_publish_bounds(p, st);
#endif
    if (p->dtype == DS_BF16) hipLaunchKernelGGL(attn_ctx_partial<bf16>, grid, dim3(256), 0, st, *p);
    else hipLaunchKernelGGL(attn_ctx_partial<float>, grid, dim3(256), 0, st, *p);
    DS_CHECK_LAUNCH("attn_ctx_partial");
    hipLaunchKernelGGL(attn_ctx_combine, dim3(p->heads, p->B, 4), dim3(256), 0, st, *p);
    DS_CHECK_LAUNCH("attn_ctx_combine");
    return DS_OK;
}

extern "C" int ds_linattn_output(const ds_attn_params* p, void* stream) {
    int rc = check(p);
    if (rc) return rc;
    DS_REQUIRE(p->out != nullptr && ds_aligned16(p->out), "linattn: bad output pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((p->N + 255) / 256, p->B);
    const size_t lds = (size_t)(p->heads * 1024 + p->heads * 32 + (p->dtype == DS_F32 ? 4 * 64 * 36 : 0)) * sizeof(float);   // + the fp32 row-transposition tiles
#if DS_BOUNDS
    linattn_publish_bounds(p, st);
#endif
    if (p->dtype == DS_BF16) hipLaunchKernelGGL(attn_out_kernel<bf16>, grid, dim3(256), lds, st, *p);
    else hipLaunchKernelGGL(attn_out_kernel<float>, grid, dim3(256), lds, st, *p);
    DS_CHECK_LAUNCH("attn_out");
    return DS_OK;
}
