// fp32 kernels of the scene encoder and the navigation fusion (K1-K5, K12 in SURVEY.md §2.2;
// reference: models/image_embedding.py:51-121, models/detr_transformer.py:170-182,
// models/nav_model.py:146-194).  Everything here stays fp32 end to end, as in the reference
// (only the LM and the heads are cast to bf16), so GEMMs use the exact-fp32
// matrix instruction v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain).  The whole encoder is
// ~2 GFLOP / 134 MB per step: latency/HBM-bound, so the kernels favour coalesced streaming of
// the weights and generality (any M/N/K, any operand layout) over MFMA peak.
#include "nv_common.h"

namespace {

// ---------------------------------------------------------------- generic fp32 GEMM
//   C[M,N] = sum_k A(m,k) * B(n,k)  (+bias[n]) ; element (m,k) of A at A[m*sam + k*sak], etc.
constexpr int FBM = 64, FBN = 64, FBK = 32, FPAD = 4;

struct GemmF32Args {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long sam, sak, sbn, sbk;
    int ldc;
    int accumulate;   // C += result
    // split-K (vector kernel only): blockIdx.z = K slice of `kchunk` (multiple of FBK); slices write raw partials to
    // part[z][M][N] and splitk_reduce_f32_kernel adds them in slice order (+bias, +C) -- deterministic, no atomics
    int ksplit, kchunk;
    float* part;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
    __shared__ float As[FBK][FBM + FPAD];
    __shared__ float Bs[FBK][FBN + FPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 2x2 waves, 32x32 each
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // loader mapping: walk the contiguous dimension with consecutive threads
    const bool a_kcontig = (p.sak == 1), b_kcontig = (p.sbk == 1);
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
#pragma unroll
        for (int e = 0; e < (FBM * FBK) / 256; ++e) {
            const int idx = e * 256 + tid;
            int r, k;
            if (a_kcontig) { k = idx % FBK; r = idx / FBK; } else { r = idx % FBM; k = idx / FBM; }
            const int gm = m0 + r, gk = k0 + k;
            As[k][r] = (gm < p.M && gk < p.K) ? p.A[gm * p.sam + gk * p.sak] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < (FBN * FBK) / 256; ++e) {
            const int idx = e * 256 + tid;
            int r, k;
            if (b_kcontig) { k = idx % FBK; r = idx / FBK; } else { r = idx % FBN; k = idx / FBN; }
            const int gn = n0 + r, gk = k0 + k;
            Bs[k][r] = (gn < p.N && gk < p.K) ? p.B[gn * p.sbn + gk * p.sbk] : 0.f;
        }
        __syncthreads();
        const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < FBK; ks += 4) {
            float fa[2], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = As[ks + kq][wm * 32 + j * 16 + i16];
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[i] = Bs[ks + kq][wn * 32 + i * 16 + i16];
            // operands swapped: D[n][m] so each lane ends with 4 consecutive n
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i], fa[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int g = lane >> 4, mi = lane & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + mi;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + i * 16 + g * 4 + r;
                if (n >= p.N) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                float* c = p.C + (long)m * p.ldc + n;
                *c = p.accumulate ? (*c + v) : v;
            }
    }
}

// Vectorised variant for the shapes that matter (weights streamed with 16-B bounds-checked buffer loads,
// next K-tile prefetched into registers under the MFMAs).  A_KC / B_KC: operand is K-contiguous.
// Preconditions (checked by the caller): K % 4 == 0 for K-contiguous operands, free dim % 4 == 0 for
// the others, leading dimensions % 4 == 0, 16-B aligned bases.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_vec_kernel(GemmF32Args p, uint32_t a_bytes, uint32_t b_bytes) {
    __shared__ __attribute__((aligned(16))) float As[FBK][FBM + FPAD];
    __shared__ __attribute__((aligned(16))) float Bs[FBK][FBN + FPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, b_bytes);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ga[2], gb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (A_KC) {
                const int k4 = (idx & 7) * 4, r = idx >> 3;
                const bool ok = (k0 + k4 < p.K);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)((((long)(m0 + r)) * p.sam + k0 + k4) * 4), 0, 0);
                ga[e] = ok ? __builtin_bit_cast(f32x4, v) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int r4 = (idx & 15) * 4, k = idx >> 4;
                const bool ok = (m0 + r4 < p.M);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)((((long)(k0 + k)) * p.sak + m0 + r4) * 4), 0, 0);
                ga[e] = ok ? __builtin_bit_cast(f32x4, v) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (B_KC) {
                const int k4 = (idx & 7) * 4, r = idx >> 3;
                const bool ok = (k0 + k4 < p.K);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)((((long)(n0 + r)) * p.sbn + k0 + k4) * 4), 0, 0);
                gb[e] = ok ? __builtin_bit_cast(f32x4, v) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int r4 = (idx & 15) * 4, k = idx >> 4;
                const bool ok = (n0 + r4 < p.N);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)((((long)(k0 + k)) * p.sbk + n0 + r4) * 4), 0, 0);
                gb[e] = ok ? __builtin_bit_cast(f32x4, v) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (A_KC) {
                const int k4 = (idx & 7) * 4, r = idx >> 3;
#pragma unroll
                for (int c = 0; c < 4; ++c) As[k4 + c][r] = ga[e][c];
            } else {
                const int r4 = (idx & 15) * 4, k = idx >> 4;
                *(f32x4*)&As[k][r4] = ga[e];
            }
            if (B_KC) {
                const int k4 = (idx & 7) * 4, r = idx >> 3;
#pragma unroll
                for (int c = 0; c < 4; ++c) Bs[k4 + c][r] = gb[e][c];
            } else {
                const int r4 = (idx & 15) * 4, k = idx >> 4;
                *(f32x4*)&Bs[k][r4] = gb[e];
            }
        }
    };
    const int kbeg = p.ksplit > 1 ? blockIdx.z * p.kchunk : 0;
    const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
    gload(kbeg);
    const int i16 = lane & 15, kq = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += FBK) {
        __syncthreads();            // previous tile's LDS reads are done
        sstore();
        __syncthreads();
        if (k0 + FBK < kend) gload(k0 + FBK);   // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < FBK; ks += 4) {
            float fa[2], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = As[ks + kq][wm * 32 + j * 16 + i16];
#pragma unroll
            for (int i = 0; i < 2; ++i) fb[i] = Bs[ks + kq][wn * 32 + i * 16 + i16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }
    const int g = lane >> 4, mi = lane & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + mi;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + i * 16 + g * 4 + r;
                if (n >= p.N) continue;
                float v = acc[i][j][r];
                if (p.ksplit > 1) { p.part[((long)blockIdx.z * p.M + m) * p.N + n] = v; continue; }
                if (p.bias) v += p.bias[n];
                float* c = p.C + (long)m * p.ldc + n;
                *c = p.accumulate ? (*c + v) : v;
            }
    }
}

// C[m,n] (+)= bias[n] + sum_z part[z][m][n], slices added in order
__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(const float* __restrict__ part, float* __restrict__ C,
                                                                const float* __restrict__ bias, int M, int N, int ldc, int S,
                                                                int accumulate) {
    const long total = (long)M * N;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += part[(long)z * total + i];
        if (bias) v += bias[n];
        float* c = C + (long)m * ldc + n;
        *c = accumulate ? (*c + v) : v;
    }
}

// ---------------------------------------------------------------- LayerNorm (one row per block)
template <int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int d,
                                                                float eps) {
    __shared__ float red[NW];
    const int m = blockIdx.x;
    const float* xr = x + (long)m * d;
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += NW * 64) s += xr[c];
    const float mean = block_sum<NW>(s, red) / (float)d;
    float v = 0.f;
    for (int c = threadIdx.x; c < d; c += NW * 64) { const float t = xr[c] - mean; v += t * t; }
    const float var = block_sum<NW>(v, red) / (float)d;
    const float rstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) { if (mean_out) mean_out[m] = mean; if (rstd_out) rstd_out[m] = rstd; }
    for (int c = threadIdx.x; c < d; c += NW * 64) y[(long)m * d + c] = (xr[c] - mean) * rstd * w[c] + b[c];
}

// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh*xh)), dxh = dy*w ; partial dgamma/dbeta per block
template <int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ w, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dx,
                                                                float* __restrict__ dg_part, float* __restrict__ db_part, int M,
                                                                int d) {
    __shared__ float red[NW];
    // column partials accumulate straight into this block's partial rows (read-modify-write by
    // the same thread each time, no races)
    float* dgp = dg_part + (long)blockIdx.x * d;
    float* dbp = db_part + (long)blockIdx.x * d;
    for (int c = threadIdx.x; c < d; c += NW * 64) { dgp[c] = 0.f; dbp[c] = 0.f; }
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const float mu = mean[m], rs = rstd[m];
        const float* xr = x + (long)m * d;
        const float* dr = dy + (long)m * d;
        float s1 = 0.f, s2 = 0.f;
        for (int c = threadIdx.x; c < d; c += NW * 64) {
            const float xh = (xr[c] - mu) * rs, dxh = dr[c] * w[c];
            s1 += dxh;
            s2 += dxh * xh;
            dgp[c] += dr[c] * xh;
            dbp[c] += dr[c];
        }
        s1 = block_sum<NW>(s1, red) / (float)d;
        s2 = block_sum<NW>(s2, red) / (float)d;
        for (int c = threadIdx.x; c < d; c += NW * 64) {
            const float xh = (xr[c] - mu) * rs, dxh = dr[c] * w[c];
            dx[(long)m * d + c] = rs * (dxh - s1 - xh * s2);
        }
    }
}

// out[c] (+)= sum_p part[p,c].  One block = 64 columns x 4 row lanes, 8 independent partial sums per thread (the first version
// walked the P rows of a column in ONE dependent chain: 36 us for 288 rows -- 33 launches per nav step, 7 ms per episode).
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int d,
                                                         long stride, int accumulate) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < d) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* q = part + c;
        int p = ty;
        for (; p + 28 < P; p += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += q[(long)(p + 4 * u) * stride];
        }
        for (; p < P; p += 4) a[0] += q[(long)p * stride];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < d) {
        s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        out[c] = accumulate ? out[c] + s : s;
    }
}

// ---------------------------------------------------------------- multi-head self-attention core
// Philox4x32-10 (counter-based RNG of the dropout kernels below and of the attention-probability dropout)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Attention-probability dropout (nn.MultiheadAttention(dropout=p), detr_transformer.py:138: torch drops the softmax output
// before P.V and rescales by 1/(1-p)).  Keep factor of probability element e = ((b*heads+hh)*N + a)*N + c: either read from an
// injected 0/1 mask [B,heads,N,N] (parity tests) or drawn from Philox counter (offset + e), word 0 -- the backward regenerates it.
struct MhaDrop {
    const float* keep;      // optional injected keep flags
    float p, inv_keep;      // p == 0: no dropout
    uint64_t seed, offset;
};
__device__ __forceinline__ float mha_keep_factor(const MhaDrop& d, long e) {
    if (d.p <= 0.f) return 1.f;
    if (d.keep) return d.keep[e] != 0.f ? d.inv_keep : 0.f;
    const uint64_t ctr = d.offset + (uint64_t)e;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)d.seed, (uint32_t)(d.seed >> 32), r);
    return ((float)(r[0] >> 8) * (1.0f / 16777216.0f)) >= d.p ? d.inv_keep : 0.f;
}
// qkv [B*N, 3h] (q|k|v, each h = heads*hd) -> out [B*N, h]; keys >= lens[b] are padding.
// One block per (b, head); P [B,heads,N,N] (the softmax output BEFORE dropout) is kept for the backward.
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                      float* __restrict__ out, float* __restrict__ P, int N, int heads, int hd,
                                                      MhaDrop drop) {
    extern __shared__ float sm[];
    float* q = sm;                 // [N][hd+1]
    float* k = q + N * (hd + 1);
    float* v = k + N * (hd + 1);
    float* s = v + N * (hd + 1);   // [N][N]
    const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
    const int h = heads * hd;
    const int len = lens[b];
    const float scale = rsqrtf((float)hd);
    for (int i = threadIdx.x; i < N * hd; i += 256) {
        const int n = i / hd, c = i % hd;
        const float* row = qkv + ((long)b * N + n) * 3 * h + hh * hd + c;
        q[n * (hd + 1) + c] = row[0] * scale;
        k[n * (hd + 1) + c] = row[h];
        v[n * (hd + 1) + c] = row[2 * h];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * N; i += 256) {
        const int a = i / N, c = i % N;
        float acc = 0.f;
        for (int e = 0; e < hd; ++e) acc += q[a * (hd + 1) + e] * k[c * (hd + 1) + e];
        s[i] = (c < len) ? acc : -INFINITY;
    }
    __syncthreads();
    // softmax per row: one wave per row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int a = wave; a < N; a += 4) {
        float mx = -INFINITY;
        for (int c = lane; c < N; c += 64) mx = fmaxf(mx, s[a * N + c]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < N; c += 64) { const float e = expf(s[a * N + c] - mx); s[a * N + c] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int c = lane; c < N; c += 64) {
            const float pv = s[a * N + c] * inv;
            const long e = (((long)b * heads + hh) * N + a) * N + c;
            P[e] = pv;
            s[a * N + c] = pv * mha_keep_factor(drop, e);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * hd; i += 256) {
        const int a = i / hd, c = i % hd;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc += s[a * N + j] * v[j * (hd + 1) + c];
        out[((long)b * N + a) * h + hh * hd + c] = acc;
    }
}

__global__ __launch_bounds__(256) void mha_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                      const float* __restrict__ dout, float* __restrict__ dqkv, int N, int heads,
                                                      int hd, MhaDrop drop) {
    extern __shared__ float sm[];
    float* q = sm;
    float* k = q + N * (hd + 1);
    float* v = k + N * (hd + 1);
    float* dO_ = v + N * (hd + 1);      // dO [N][hd+1]
    float* pp = dO_ + N * (hd + 1);     // P  [N][N]
    float* ds = pp + N * N;            // dS [N][N]
    const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
    const int h = heads * hd;
    const float scale = rsqrtf((float)hd);
    for (int i = threadIdx.x; i < N * hd; i += 256) {
        const int n = i / hd, c = i % hd;
        const float* row = qkv + ((long)b * N + n) * 3 * h + hh * hd + c;
        q[n * (hd + 1) + c] = row[0];
        k[n * (hd + 1) + c] = row[h];
        v[n * (hd + 1) + c] = row[2 * h];
        dO_[n * (hd + 1) + c] = dout[((long)b * N + n) * h + hh * hd + c];
    }
    for (int i = threadIdx.x; i < N * N; i += 256) pp[i] = P[((long)b * heads + hh) * N * N + i];
    __syncthreads();
    // dP = dO V^T
    for (int i = threadIdx.x; i < N * N; i += 256) {
        const int a = i / N, c = i % N;
        float acc = 0.f;
        for (int e = 0; e < hd; ++e) acc += dO_[a * (hd + 1) + e] * v[c * (hd + 1) + e];
        ds[i] = acc;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // with dropout: out = (P*m) V, m = keep/(1-p)  ->  dP = m * (dO V^T), dS = P * (dP - sum_c dP*P), dV = (P*m)^T dO.
    // ds holds dP, then dS; pp holds P, then P*m (P itself is not needed once dS is formed).
    for (int a = wave; a < N; a += 4) {
        float dot = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float m = mha_keep_factor(drop, (((long)b * heads + hh) * N + a) * N + c);
            const float dp = ds[a * N + c] * m;
            ds[a * N + c] = dp;
            dot += dp * pp[a * N + c];
        }
        dot = wave_sum(dot);
        for (int c = lane; c < N; c += 64) {
            const float pv = pp[a * N + c];
            ds[a * N + c] = pv * (ds[a * N + c] - dot);
            if (drop.p > 0.f) pp[a * N + c] = pv * mha_keep_factor(drop, (((long)b * heads + hh) * N + a) * N + c);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * hd; i += 256) {
        const int a = i / hd, c = i % hd;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < N; ++j) {
            dq += ds[a * N + j] * k[j * (hd + 1) + c];
            dk += ds[j * N + a] * q[j * (hd + 1) + c];
            dv += pp[j * N + a] * dO_[j * (hd + 1) + c];
        }
        float* row = dqkv + ((long)b * N + a) * 3 * h + hh * hd + c;
        row[0] = dq * scale;
        row[h] = dk * scale;
        row[2 * h] = dv;
    }
}

// ---------------------------------------------------------------- elementwise / row ops
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}
// out = a + b   (b may be broadcast over rows when b_rows == 1)
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n, int d, int b_bcast) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
        out[i] = a[i] + (b_bcast ? b[i % d] : b[i]);
}
// out = a * b (elementwise; dropout masks pre-scaled by 1/(1-p))
__global__ __launch_bounds__(256) void mul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = a[i] * b[i];
}
// nn.Dropout with the keep mask drawn IN the kernel (Philox4x32-10, counter = element group, key = seed): one launch instead of
// torch.rand + compare + cast + scale + multiply, and nothing to store for the backward -- it calls the same kernel with the same
// (seed, offset) on the incoming gradient and regenerates the identical mask.  (nav_model.py:91,99-102 drop_env p=0.4;
// image_embedding.py:73-74 and detr_transformer.py:170-182 p=0.1.)  The reference's mask stream is torch's CUDA Philox
// generator: a different, equally valid stream (dropout cannot be bit-matched across devices anyway).
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float p,
                                                          float scale, uint64_t seed, uint64_t offset) {
    const long groups = (n + 3) / 4;
    for (long gi = blockIdx.x * 256L + threadIdx.x; gi < groups; gi += gridDim.x * 256L) {
        const uint64_t ctr = offset + (uint64_t)gi;
        uint32_t r[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const long i = gi * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j < n) {
                const float u = (float)(r[j] >> 8) * (1.0f / 16777216.0f);     // uniform on [0, 1), 24 bits
                out[i + j] = u >= p ? x[i + j] * scale : 0.f;
            }
    }
}
// out[m,:] = x[m,:] * s[m]
__global__ __launch_bounds__(256) void rowscale_f32_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           float* __restrict__ out, long n, int d) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = x[i] * s[i / d];
}
// out[i,:] = (idx[i] >= 0 ? src[idx[i],:] : 0) (+ base[i,:] if base)
__global__ __launch_bounds__(256) void gather_add_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                             const float* __restrict__ base, float* __restrict__ out, long n,
                                                             int d) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const long r = i / d;
        const int c = (int)(i % d);
        const int s = idx[r];
        float v = s >= 0 ? src[(long)s * d + c] : 0.f;
        if (base) v += base[i];
        out[i] = v;
    }
}
// dst[r,:] = sum_{i : idx[i]==r} src[i,:]   for r in [0,R): deterministic "index_add" (small R or small n)
__global__ __launch_bounds__(256) void index_sum_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                            float* __restrict__ dst, int n, int R, int d, int accumulate) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i)
        if (idx[i] == r) s += src[(long)i * d + c];
    float* o = dst + (long)r * d + c;
    *o = accumulate ? *o + s : s;
}
// pooled[b,:] = sum_n x[b,n,:]*mask[b,n] / sum_n mask[b,n]       (mp3d_agent.py:685-686)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          float* __restrict__ out, int N, int d) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float s = 0.f, cnt = 0.f;
    for (int n = 0; n < N; ++n) {
        const float mk = mask[b * N + n];
        s += x[((long)b * N + n) * d + c] * mk;
        cnt += mk;
    }
    out[(long)b * d + c] = s / cnt;
}

inline int grid_for(long total, int cap = 256 * 8) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" {

// layout 0: NT  A[M,K] B[N,K] ; 1: NN  A[M,K] B[K,N] ; 2: TN  A[K,M] B[K,N]
size_t nv_gemm_f32_workspace_bytes(int M, int N) { return (size_t)8 * M * N * sizeof(float); }

int nv_gemm_f32_ws(int layout, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                   int ldc, int accumulate, void* workspace, void* stream);

int nv_gemm_f32(int layout, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                int ldc, int accumulate, void* stream) {
    return nv_gemm_f32_ws(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, accumulate, nullptr, stream);
}

// with a workspace of nv_gemm_f32_workspace_bytes(M, N): launches that would leave most CUs idle (the encoder's
// [288 x 1024] outputs are 80 tiles of 64x64) are cut into up to 8 K slices, reduced in slice order by a second kernel
int nv_gemm_f32_ws(int layout, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                   int ldc, int accumulate, void* workspace, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    GemmF32Args p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.ldc = ldc; p.accumulate = accumulate;
    p.ksplit = 1; p.kchunk = K; p.part = nullptr;
    switch (layout) {
        case 0: p.sam = lda; p.sak = 1; p.sbn = ldb; p.sbk = 1; break;
        case 1: p.sam = lda; p.sak = 1; p.sbn = 1; p.sbk = ldb; break;
        case 2: p.sam = 1; p.sak = lda; p.sbn = 1; p.sbk = ldb; break;
        default: return NV_ERR_ARG;
    }
    const dim3 grid((N + FBN - 1) / FBN, (M + FBM - 1) / FBM);
    // vector path: 16-B loads need aligned, 4-divisible extents along each operand's contiguous dimension
    const bool a_kc = (layout != 2), b_kc = (layout == 0);
    const bool aligned = !((((uintptr_t)A) | ((uintptr_t)B)) & 15) && !(lda & 3) && !(ldb & 3) &&
                         (a_kc ? !(K & 3) : !(M & 3)) && (b_kc ? !(K & 3) : !(N & 3));
    if (aligned) {
        const long a_rows = a_kc ? M : K, a_cols = a_kc ? K : M, b_rows = b_kc ? N : K, b_cols = b_kc ? K : N;
        const uint32_t ab = (uint32_t)(((a_rows - 1) * (long)lda + a_cols) * 4), bb = (uint32_t)(((b_rows - 1) * (long)ldb + b_cols) * 4);
        hipStream_t st = (hipStream_t)stream;
        dim3 g3 = grid;
        const int blocks = grid.x * grid.y, ktiles = (K + FBK - 1) / FBK;
        if (workspace && blocks < 512 && ktiles >= 8) {
            // aim at ~4 blocks per CU: these launches are latency-bound (one 64x64x32 step per barrier pair)
            int S = (1024 + blocks - 1) / blocks;
            if (S > 8) S = 8;
            if (S > ktiles / 4) S = ktiles / 4;
            if (S >= 2) {
                p.kchunk = ((ktiles + S - 1) / S) * FBK;
                p.ksplit = (K + p.kchunk - 1) / p.kchunk;
                p.part = (float*)workspace;
                g3.z = p.ksplit;
            }
        }
        if (layout == 0) NV_LAUNCH((gemm_f32_vec_kernel<true, true>), g3, dim3(256), 0, st, p, ab, bb);
        else if (layout == 1) NV_LAUNCH((gemm_f32_vec_kernel<true, false>), g3, dim3(256), 0, st, p, ab, bb);
        else NV_LAUNCH((gemm_f32_vec_kernel<false, false>), g3, dim3(256), 0, st, p, ab, bb);
        if (p.ksplit > 1)
            NV_LAUNCH(splitk_reduce_f32_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (const float*)p.part, C, bias, M, N, ldc,
                      p.ksplit, accumulate);
        return nv_check_launch();
    }
    NV_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return nv_check_launch();
}

int nv_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int M, int d,
                         float eps, void* stream) {
    if (!x || !w || !b || !y) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    NV_LAUNCH(layernorm_fwd_kernel<4>, dim3(M), dim3(256), 0, (hipStream_t)stream, x, w, b, y, mean, rstd, d, eps);
    return nv_check_launch();
}

size_t nv_layernorm_bwd_workspace_bytes(int d) { return (size_t)2 * 128 * d * sizeof(float); }

// dgamma/dbeta are ACCUMULATED into gw/gb (autograd-free use) when accumulate != 0
int nv_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                         float* gw, float* gb, void* workspace, int M, int d, int accumulate, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx || !gw || !gb || !workspace) return NV_ERR_ARG;
    if (M == 0) return NV_OK;
    const int P = M < 128 ? M : 128;
    float* dgp = (float*)workspace;
    float* dbp = dgp + (size_t)128 * d;
    hipStream_t st = (hipStream_t)stream;
    NV_LAUNCH(layernorm_bwd_kernel<4>, dim3(P), dim3(256), 0, st, dy, x, w, mean, rstd, dx, dgp, dbp, M, d);
    NV_LAUNCH(colsum_f32_kernel, dim3((d + 63) / 64), dim3(256), 0, st, dgp, gw, P, d, (long)d, accumulate);
    NV_LAUNCH(colsum_f32_kernel, dim3((d + 63) / 64), dim3(256), 0, st, dbp, gb, P, d, (long)d, accumulate);
    return nv_check_launch();
}

// out[c] (+)= sum_m x[m,c]   (bias gradients)
int nv_colsum_f32(const float* x, float* out, int M, int d, int ld, int accumulate, void* stream) {
    if (!x || !out) return NV_ERR_ARG;
    NV_LAUNCH(colsum_f32_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, out, M, d, (long)ld,
                       accumulate);
    return nv_check_launch();
}

static size_t mha_lds(int N, int hd, int bwd) {
    return (size_t)((bwd ? 4 : 3) * N * (hd + 1) + (bwd ? 2 : 1) * N * N) * sizeof(float);
}

static bool mha_drop_args(MhaDrop& d, const float* keep, float p, unsigned long long seed, unsigned long long offset) {
    if (!(p >= 0.f) || !(p < 1.f)) return false;
    d.keep = keep; d.p = p; d.inv_keep = 1.f / (1.f - p); d.seed = (uint64_t)seed; d.offset = (uint64_t)offset;
    return true;
}

// `keep` (optional, [B,heads,N,N] 0/1 flags) overrides the Philox draw; p == 0 is nv_mha_fwd_f32.  P receives the
// probabilities BEFORE dropout; nv_mha_bwd_drop_f32 must be given the same (keep, p, seed, offset).
int nv_mha_fwd_drop_f32(const float* qkv, const int* lens, float* out, float* P, const float* keep, float p,
                        unsigned long long seed, unsigned long long offset, int B, int N, int heads, int hd, void* stream) {
    if (!qkv || !lens || !out || !P) return NV_ERR_ARG;
    MhaDrop drop;
    if (!mha_drop_args(drop, keep, p, seed, offset)) return NV_ERR_ARG;
    if (B == 0 || N == 0) return NV_OK;
    const size_t lds = mha_lds(N, hd, 0);
    if (lds > 160 * 1024) return NV_ERR_SHAPE;
    if (hipFuncSetAttribute((const void*)mha_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return NV_ERR_LAUNCH;
    NV_LAUNCH(mha_fwd_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, lens, out, P, N, heads, hd, drop);
    return nv_check_launch();
}
int nv_mha_fwd_f32(const float* qkv, const int* lens, float* out, float* P, int B, int N, int heads, int hd, void* stream) {
    return nv_mha_fwd_drop_f32(qkv, lens, out, P, nullptr, 0.f, 0ull, 0ull, B, N, heads, hd, stream);
}

int nv_mha_bwd_drop_f32(const float* qkv, const float* P, const float* dout, float* dqkv, const float* keep, float p,
                        unsigned long long seed, unsigned long long offset, int B, int N, int heads, int hd, void* stream) {
    if (!qkv || !P || !dout || !dqkv) return NV_ERR_ARG;
    MhaDrop drop;
    if (!mha_drop_args(drop, keep, p, seed, offset)) return NV_ERR_ARG;
    if (B == 0 || N == 0) return NV_OK;
    const size_t lds = mha_lds(N, hd, 1);
    if (lds > 160 * 1024) return NV_ERR_SHAPE;
    if (hipFuncSetAttribute((const void*)mha_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return NV_ERR_LAUNCH;
    NV_LAUNCH(mha_bwd_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, P, dout, dqkv, N, heads, hd, drop);
    return nv_check_launch();
}
int nv_mha_bwd_f32(const float* qkv, const float* P, const float* dout, float* dqkv, int B, int N, int heads, int hd,
                   void* stream) {
    return nv_mha_bwd_drop_f32(qkv, P, dout, dqkv, nullptr, 0.f, 0ull, 0ull, B, N, heads, hd, stream);
}

int nv_gelu_fwd_f32(const float* x, float* y, long n, void* stream) {
    if (!x || !y) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return nv_check_launch();
}
int nv_gelu_bwd_f32(const float* x, const float* dy, float* dx, long n, void* stream) {
    if (!x || !dy || !dx) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(gelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
    return nv_check_launch();
}
int nv_add_f32(const float* a, const float* b, float* out, long n, int d, int b_bcast, void* stream) {
    if (!a || !b || !out) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, d, b_bcast);
    return nv_check_launch();
}
int nv_mul_f32(const float* a, const float* b, float* out, long n, void* stream) {
    if (!a || !b || !out) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(mul_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return nv_check_launch();
}
int nv_dropout_f32(const float* x, float* out, long n, float p, unsigned long long seed, unsigned long long offset, void* stream) {
    if (!x || !out || !(p >= 0.f) || !(p < 1.f)) return NV_ERR_ARG;
    if (n == 0) return NV_OK;
    NV_LAUNCH(dropout_f32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, n, p, 1.f / (1.f - p),
              (uint64_t)seed, (uint64_t)offset);
    return nv_check_launch();
}
int nv_rowscale_f32(const float* x, const float* s, float* out, long rows, int d, void* stream) {
    if (!x || !s || !out) return NV_ERR_ARG;
    if (rows == 0) return NV_OK;
    NV_LAUNCH(rowscale_f32_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, x, s, out, rows * d, d);
    return nv_check_launch();
}
int nv_gather_add_f32(const float* src, const int* idx, const float* base, float* out, long rows, int d, void* stream) {
    if (!src || !idx || !out) return NV_ERR_ARG;
    if (rows == 0) return NV_OK;
    NV_LAUNCH(gather_add_f32_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, src, idx, base, out,
                       rows * d, d);
    return nv_check_launch();
}
int nv_index_sum_f32(const float* src, const int* idx, float* dst, int n, int R, int d, int accumulate, void* stream) {
    if (!src || !idx || !dst) return NV_ERR_ARG;
    if (R == 0) return NV_OK;
    NV_LAUNCH(index_sum_f32_kernel, dim3((d + 255) / 256, R), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, R,
                       d, accumulate);
    return nv_check_launch();
}
int nv_masked_mean_f32(const float* x, const float* mask, float* out, int B, int N, int d, void* stream) {
    if (!x || !mask || !out) return NV_ERR_ARG;
    if (B == 0) return NV_OK;
    NV_LAUNCH(masked_mean_kernel, dim3((d + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, mask, out, N, d);
    return nv_check_launch();
}

}  // extern "C"
