// Skinny NT GEMM for the decode steps of generation: C[M <= 16, N] = A[M,K] @ W[N,K]^T in bf16,
// fp32 accumulate, optional residual.  One greedy-decoding step multiplies B (= 8) token rows with every weight matrix
// of the LM: 13.5 GB of weights per step at 7B, so the kernel is a weight STREAMER -- the 256x256 / 128x128 tile GEMMs
// reach ~1 TB/s here (a few dozen blocks, each re-reading a mostly empty A tile).
//
// Mapping: block = 16 output columns (one MFMA tile) x the whole K, 8 waves each owning a contiguous K slice; a wave
// keeps UNROLL x 16 B per lane of W in flight straight from HBM into the MFMA B-operand layout (lane (n, kg) holds
// W[n0+n][k + kg*8 .. +8]: no LDS on the way), x rows come from L2 into the A-operand layout (rows >= M are zero).
// v_mfma_f32_16x16x32_bf16 does the 16x16x32 products; the 8 partial tiles meet in LDS and wave 0 writes the M rows.
// HBM-bound: bytes = N*K*2 per launch; N/16 blocks (256 at N=4096, 1376 at N=22016).
#include "nv_common.h"
#include <stdlib.h>

namespace {

constexpr int GV_WAVES = 8;

// Round 2: NTILE column tiles per block.  With one tile per block the x rows (L2-resident) are re-read by every 16-column
// block, so the vector-memory path carries 2 bytes for every weight byte and saturates at ~9 TB/s -- 4.4 TB/s of weights
// (tools/gemv_fp8_probe.py).  A block that owns NTILE tiles loads each x fragment once and feeds NTILE MFMAs from it:
// (NTILE+1)/NTILE bytes through L1 per weight byte.  Measured: the gain is real but modest (the x traffic was not the whole
// story): +9..+15 % at the 13B shapes, nothing at K = 4096; NTILE is chosen on the host from those measurements.
template <bool RESID, int NTILE>
__global__ __launch_bounds__(GV_WAVES * 64) void gemv_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                  bf16_t* __restrict__ C, const bf16_t* __restrict__ R, int M, int N,
                                                                  int K, int lda, int ldw, int ldc, int ldr) {
    constexpr int UNROLL = 8 / NTILE;
    __shared__ float part[GV_WAVES][NTILE * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16 * NTILE;
    const int steps = K / 32;                                     // K % 32 == 0 (checked on the host)
    const int per = (steps + GV_WAVES - 1) / GV_WAVES;
    const int s_beg = wave * per, s_end = min(steps, s_beg + per);
    const bool m_ok = idx < M;
    const bf16_t* wp[NTILE];
    bool n_ok[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const int n = n0 + t * 16 + idx;
        n_ok[t] = n < N;
        wp[t] = W + (long)(n_ok[t] ? n : N - 1) * ldw + kg * 8;
    }
    const bf16_t* ap = A + (long)(m_ok ? idx : 0) * lda + kg * 8;
    f32x4 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {};
    int s = s_beg;
    for (; s + UNROLL <= s_end; s += UNROLL) {
        bf16x8 wf[UNROLL][NTILE], af[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NTILE; ++t) wf[u][t] = *(const bf16x8*)(wp[t] + (long)(s + u) * 32);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) af[u] = m_ok ? *(const bf16x8*)(ap + (long)(s + u) * 32) : zero;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NTILE; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(n_ok[t] ? wf[u][t] : zero, af[u], acc[t], 0, 0, 0);   // D[n][m]
    }
    for (; s < s_end; ++s) {
        const bf16x8 af = m_ok ? *(const bf16x8*)(ap + (long)s * 32) : zero;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const bf16x8 wf = *(const bf16x8*)(wp[t] + (long)s * 32);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(n_ok[t] ? wf : zero, af, acc[t], 0, 0, 0);
        }
    }
    // D layout: lane holds D[row = kg*4 + r][col = idx] with rows = n (first operand), cols = m (second operand)
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][t * 16 + kg * 4 + r][idx] = acc[t][r];
    __syncthreads();
    // NTILE*256 outputs (16 m x NTILE*16 n) over the block's 512 threads
    for (int e = tid; e < NTILE * 256; e += GV_WAVES * 64) {
        const int m = e / (NTILE * 16), nn = e % (NTILE * 16);
        if (m < M && n0 + nn < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < GV_WAVES; ++w) v += part[w][nn][m];
            if (RESID) v = bf2f(R[(long)m * ldr + n0 + nn]) + rbf(v);     // torch: resid + bf16(x W^T)
            C[(long)m * ldc + n0 + nn] = f2bf(v);
        }
    }
}

inline int gemv_ntile(int N, int K) {
    static const int forced = [] { const char* e = getenv("NV_GEMV_NTILE"); return e ? atoi(e) : 0; }();   // measurement knob
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // measured (tools/gemv_fp8_probe.py, profiles/r02_gemv_probe.txt): two tiles per block pay when the K loop is long enough
    // (13B shapes: +9..+15 %) and the grid keeps >= 160 blocks; at K = 4096 (7B) one tile per block stays best
    return (K >= 5120 && N >= 5120) ? 2 : 1;
}

}  // namespace

extern "C" {

// C[M,N] = A[M,K] @ W[N,K]^T (+ R), M <= 16, K % 32 == 0, 16-B aligned rows.  epilogue: 0 store, 2 residual (as nv_gemm_bf16).
int nv_gemv_bf16(const void* A, const void* W, void* C, const void* R, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                 int epilogue, void* stream) {
    if (!A || !W || !C || M < 0 || N < 0 || K < 0) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    if (M > 16 || (K & 31) || (lda & 7) || (ldw & 7) || ((((uintptr_t)A) | ((uintptr_t)W)) & 15)) return NV_ERR_SHAPE;
    if (epilogue != 0 && epilogue != 2) return NV_ERR_ARG;
    if (epilogue == 2 && !R) return NV_ERR_ARG;
    {   // the full-line streamer (gemv_stream.hip) takes every shape of its fast path
        const int rc = nvi_gemv_stream(A, W, nullptr, C, R, M, N, K, lda, ldw, ldc, ldr, epilogue == 2, 0, stream);
        if (rc != NV_ERR_SHAPE) return rc;
    }
    const int nt = gemv_ntile(N, K);
    const dim3 grid((N + 16 * nt - 1) / (16 * nt)), block(GV_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
#define NV_GV(RES, NT) NV_LAUNCH((gemv_bf16_kernel<RES, NT>), grid, block, 0, st, (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, \
                                 (const bf16_t*)(RES ? R : nullptr), M, N, K, lda, ldw, ldc, ldr)
    if (epilogue == 2) { if (nt == 4) NV_GV(true, 4); else if (nt == 2) NV_GV(true, 2); else NV_GV(true, 1); }
    else { if (nt == 4) NV_GV(false, 4); else if (nt == 2) NV_GV(false, 2); else NV_GV(false, 1); }
#undef NV_GV
    return nv_check_launch();
}

}  // extern "C"
