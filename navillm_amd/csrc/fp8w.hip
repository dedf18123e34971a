// Weight-only fp8 (OCP e4m3fn) for the LM's Linear layers (BASELINE config 5: Vicuna-13B, inference).
//
//   W[n, :]  ~=  s[n] * q[n, :],   s[n] = max_k |W[n,k]| / 448 (fp32, one per OUTPUT channel),  q = e4m3fn(W / s) (RNE)
//
// The reference has no quantised path; the semantics pinned by the fixtures (tests/golden/g11_fp8_*.npz) are "the reference
// model run on the de-quantised weights bf16(s*q)".  Three kernels:
//   * nv_fp8_quant_rows      bf16 [N,K] -> u8 [N,K] + fp32 scales     (one-off, at load time; exact RNE in integer math)
//   * nv_fp8_dequant_rows    u8 + scales -> bf16 [N,K] = bf16(s*q)    (HBM-bound pre-pass in front of the prefill GEMMs: the
//                            layer's weights live in HBM as fp8 only; one bf16 scratch panel is reused by every GEMM)
//   * nv_gemv_fp8w           C[M<=16, N] = A @ bf16(s*q)^T (+R)       (the decode steps: a weight STREAMER, so fp8 halves the
//                            bytes per generated token; 16 fp8 per lane per load, converted in registers with
//                            v_cvt_pk_f32_fp8 -- gfx950 decodes OCP e4m3fn -- then v_cvt_pk_bf16_f32)
// nv_fp8_decode_table exposes the in-register decode for all 256 codes so a test can pin it against torch.float8_e4m3fn.
#include "nv_common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float* f) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}

// ---- e4m3fn decode: 4 codes of one dword -> 4 floats (hardware conversion; exactness checked by tests/test_fp8_gpu.py)
__device__ __forceinline__ void fp8x4_to_f32(uint32_t w, float* f) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
    const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}

// ---- fp32 -> e4m3fn, round to nearest even, |x| <= 464 (the quantiser guarantees |x| <= 448(1+eps)); integer math
__device__ __forceinline__ uint32_t f32_to_e4m3fn(float x) {
    const uint32_t b = __float_as_uint(x);
    const uint32_t sign = (b >> 24) & 0x80u;
    const float a = fabsf(x);
    uint32_t code;
    if (a >= 0.015625f) {                                  // >= 2^-6: normal in e4m3
        uint32_t m = b & 0x7fffffffu;
        m += 0x7ffffu + ((m >> 20) & 1u);                  // RNE at bit 20 (keep 3 mantissa bits)
        code = (m >> 20) - ((127u - 7u) << 3);
    } else {
        code = (uint32_t)rintf(a * 512.f);                 // subnormal grid 2^-9; 8 rolls over into the first normal
    }
    if (code > 0x7eu) code = 0x7eu;                        // saturate at 448 (0x7f is NaN)
    return sign | code;
}

// one block per row: amax -> scale -> codes.  K % 8 == 0, 16-B aligned rows.
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const bf16_t* __restrict__ W, uint8_t* __restrict__ Q, float* __restrict__ S,
                                                             int K, int ldw, int ldq) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const bf16_t* w = W + (long)n * ldw;
    float amax = 0.f;
    for (int k = tid * 8; k < K; k += 256 * 8) {
        float f[8];
        ld8(w + k, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = amax > 0.f ? __fdiv_rn(amax, 448.f) : 1.f;
    if (tid == 0) S[n] = s;
    uint8_t* q = Q + (long)n * ldq;
    for (int k = tid * 8; k < K; k += 256 * 8) {
        float f[8];
        ld8(w + k, f);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo |= f32_to_e4m3fn(__fdiv_rn(f[j], s)) << (8 * j);
            hi |= f32_to_e4m3fn(__fdiv_rn(f[4 + j], s)) << (8 * j);
        }
        *(u32x2*)(q + k) = u32x2{lo, hi};
    }
}

// out[n,k] = bf16(s[n] * q[n,k]); 16 codes (16 B) per thread per step.  K % 16 == 0.
__global__ __launch_bounds__(256) void fp8_dequant_rows_kernel(const uint8_t* __restrict__ Q, const float* __restrict__ S,
                                                               bf16_t* __restrict__ out, int N, int K, int ldq, int ldo) {
    const int per_row = K / 16;
    const long total = (long)N * per_row;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int n = (int)(i / per_row), c = (int)(i % per_row);
        const u32x4 w = *(const u32x4*)(Q + (long)n * ldq + c * 16);
        const float s = S[n];
        u32x4 o0, o1;
        float f[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) fp8x4_to_f32(w[j], f + 4 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = pack2bf(f[2 * j] * s, f[2 * j + 1] * s);
            o1[j] = pack2bf(f[8 + 2 * j] * s, f[8 + 2 * j + 1] * s);
        }
        bf16_t* op = out + (long)n * ldo + c * 16;
        *(u32x4*)op = o0;
        *(u32x4*)(op + 8) = o1;
    }
}

__global__ void fp8_decode_table_kernel(bf16_t* __restrict__ out) {
    const int t = threadIdx.x;                             // 64 threads x 4 codes
    float f[4];
    const uint32_t w = (uint32_t)(4 * t) | ((uint32_t)(4 * t + 1) << 8) | ((uint32_t)(4 * t + 2) << 16) | ((uint32_t)(4 * t + 3) << 24);
    fp8x4_to_f32(w, f);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * t + j] = f2bf(f[j]);
}

// ---- decode-step GEMV with fp8 weights.  Same mapping as gemv_bf16_kernel (block = 16 output columns x all of K, 8 waves on
// K slices, partial tiles meet in LDS) but a lane streams 16 B = 16 codes per load: lane (n, kg) holds W[n][k0 + kg*16 .. +16)
// of a 64-deep step and feeds TWO MFMAs (its first / second 8 codes); the x operand is loaded with the same k permutation.
// Semantics = the de-quantised weight bf16(s[n] * q[n,k]) exactly as nv_fp8_dequant_rows writes it (lane (n, kg) always works on row
// n, so s[n] is a per-lane constant multiplied in before the bf16 pack): decode and prefill see the same weights.
constexpr int GF_WAVES = 8;

// NTILE column tiles per block share the x fragments (see gemv_bf16.hip): with one tile per block three 16-B loads (one of
// codes, two of x) go through the vector-memory path per 16 B of codes and the streamer stalls at ~3 TB/s of codes; with NTILE
// tiles it is (NTILE + 2) / NTILE.
template <bool RESID, int NTILE>
__global__ __launch_bounds__(GF_WAVES * 64) void gemv_fp8w_kernel(const bf16_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                                  const float* __restrict__ S, bf16_t* __restrict__ C,
                                                                  const bf16_t* __restrict__ R, int M, int N, int K, int lda, int ldw,
                                                                  int ldc, int ldr) {
    constexpr int UNROLL = NTILE >= 2 ? 2 : 4;
    __shared__ float part[GF_WAVES][NTILE * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16 * NTILE;
    const int steps = K / 64;                                     // K % 64 == 0 (checked on the host)
    const int per = (steps + GF_WAVES - 1) / GF_WAVES;
    const int s_beg = wave * per, s_end = min(steps, s_beg + per);
    const bool m_ok = idx < M;
    const uint8_t* wp[NTILE];
    bool n_ok[NTILE];
    float sc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const int n = n0 + t * 16 + idx;
        n_ok[t] = n < N;
        wp[t] = W + (long)(n_ok[t] ? n : N - 1) * ldw + kg * 16;
        sc[t] = n_ok[t] ? S[n] : 0.f;
    }
    const bf16_t* ap = A + (long)(m_ok ? idx : 0) * lda + kg * 16;
    f32x4 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {};
    auto fma_tile = [&](int t, const u32x4& w, const bf16x8& a0, const bf16x8& a1) {
        float f[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) fp8x4_to_f32(w[j], f + 4 * j);
        u32x4 b0, b1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b0[j] = pack2bf(f[2 * j] * sc[t], f[2 * j + 1] * sc[t]);
            b1[j] = pack2bf(f[8 + 2 * j] * sc[t], f[8 + 2 * j + 1] * sc[t]);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b0), a0, acc[t], 0, 0, 0);   // D[n][m]
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b1), a1, acc[t], 0, 0, 0);
    };
    int s = s_beg;
    for (; s + UNROLL <= s_end; s += UNROLL) {
        u32x4 wf[UNROLL][NTILE];
        bf16x8 a0[UNROLL], a1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NTILE; ++t) wf[u][t] = n_ok[t] ? *(const u32x4*)(wp[t] + (long)(s + u) * 64) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            a0[u] = m_ok ? *(const bf16x8*)(ap + (long)(s + u) * 64) : zero;
            a1[u] = m_ok ? *(const bf16x8*)(ap + (long)(s + u) * 64 + 8) : zero;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NTILE; ++t) fma_tile(t, wf[u][t], a0[u], a1[u]);
    }
    for (; s < s_end; ++s) {
        const bf16x8 a0 = m_ok ? *(const bf16x8*)(ap + (long)s * 64) : zero;
        const bf16x8 a1 = m_ok ? *(const bf16x8*)(ap + (long)s * 64 + 8) : zero;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) fma_tile(t, n_ok[t] ? *(const u32x4*)(wp[t] + (long)s * 64) : u32x4{0, 0, 0, 0}, a0, a1);
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][t * 16 + kg * 4 + r][idx] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < NTILE * 256; e += GF_WAVES * 64) {
        const int m = e / (NTILE * 16), nn = e % (NTILE * 16);
        if (m < M && n0 + nn < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < GF_WAVES; ++w) v += part[w][nn][m];
            if (RESID) v = bf2f(R[(long)m * ldr + n0 + nn]) + rbf(v);     // torch: resid + bf16(x W^T)
            C[(long)m * ldc + n0 + nn] = f2bf(v);
        }
    }
}

inline int grid_cap(long total) {
    long b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > 256 * 16 ? 256 * 16 : b);
}

}  // namespace

extern "C" {

int nv_fp8_quant_rows(const void* W, void* Q, float* scales, int N, int K, int ldw, int ldq, void* stream) {
    if (!W || !Q || !scales || N < 0 || K < 0) return NV_ERR_ARG;
    if ((K & 7) || (ldw & 7) || (ldq & 7) || ((uintptr_t)W & 15) || ((uintptr_t)Q & 7)) return NV_ERR_SHAPE;
    if (N == 0 || K == 0) return NV_OK;
    NV_LAUNCH(fp8_quant_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, (uint8_t*)Q, scales, K, ldw, ldq);
    return nv_check_launch();
}

int nv_fp8_dequant_rows(const void* Q, const float* scales, void* out, int N, int K, int ldq, int ldo, void* stream) {
    if (!Q || !scales || !out || N < 0 || K < 0) return NV_ERR_ARG;
    if ((K & 15) || (ldq & 15) || (ldo & 7) || ((uintptr_t)Q & 15) || ((uintptr_t)out & 15)) return NV_ERR_SHAPE;
    if (N == 0 || K == 0) return NV_OK;
    NV_LAUNCH(fp8_dequant_rows_kernel, dim3(grid_cap((long)N * K / 16)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)Q, scales,
              (bf16_t*)out, N, K, ldq, ldo);
    return nv_check_launch();
}

int nv_fp8_decode_table(void* out256_bf16, void* stream) {
    if (!out256_bf16) return NV_ERR_ARG;
    NV_LAUNCH(fp8_decode_table_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (bf16_t*)out256_bf16);
    return nv_check_launch();
}

// C[M,N] = A[M,K] @ bf16(scales[n] * Wq[N,K])^T (+ R), M <= 16, K % 64 == 0; epilogue 0 store, 2 residual (as nv_gemv_bf16)
int nv_gemv_fp8w(const void* A, const void* Wq, const float* scales, void* C, const void* R, int M, int N, int K, int lda, int ldw,
                 int ldc, int ldr, int epilogue, void* stream) {
    if (!A || !Wq || !scales || !C || M < 0 || N < 0 || K < 0) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    if (M > 16 || (K & 63) || (lda & 7) || (ldw & 15) || ((((uintptr_t)A) | ((uintptr_t)Wq)) & 15)) return NV_ERR_SHAPE;
    if (epilogue != 0 && epilogue != 2) return NV_ERR_ARG;
    if (epilogue == 2 && !R) return NV_ERR_ARG;
    {   // the full-line streamer (gemv_stream.hip) takes every shape of its fast path
        const int rc = nvi_gemv_stream(A, Wq, scales, C, R, M, N, K, lda, ldw, ldc, ldr, epilogue == 2, 1, stream);
        if (rc != NV_ERR_SHAPE) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    static const int forced = [] { const char* e = getenv("NV_GEMV_NTILE"); return e ? atoi(e) : 0; }();   // measurement knob
    // measured (profiles/r02_gemv_probe.txt): two tiles per block win by 6..27 % whenever the grid keeps >= 160 blocks
    const int nt = (forced == 1 || forced == 2 || forced == 4) ? forced : (N >= 5120 ? 2 : 1);
    const dim3 grid((N + 16 * nt - 1) / (16 * nt)), block(GF_WAVES * 64);
#define NV_GF(RES, NT) NV_LAUNCH((gemv_fp8w_kernel<RES, NT>), grid, block, 0, st, (const bf16_t*)A, (const uint8_t*)Wq, scales, (bf16_t*)C, \
                                 (const bf16_t*)(RES ? R : nullptr), M, N, K, lda, ldw, ldc, ldr)
    if (epilogue == 2) { if (nt == 4) NV_GF(true, 4); else if (nt == 2) NV_GF(true, 2); else NV_GF(true, 1); }
    else { if (nt == 4) NV_GF(false, 4); else if (nt == 2) NV_GF(false, 2); else NV_GF(false, 1); }
#undef NV_GF
    return nv_check_launch();
}

}  // extern "C"
