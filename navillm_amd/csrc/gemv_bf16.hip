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

namespace {

constexpr int GV_WAVES = 8;
constexpr int GV_UNROLL = 8;

template <bool RESID>
__global__ __launch_bounds__(GV_WAVES * 64) void gemv_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                  bf16_t* __restrict__ C, const bf16_t* __restrict__ R, int M, int N,
                                                                  int K, int lda, int ldw, int ldc, int ldr) {
    __shared__ float part[GV_WAVES][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int steps = K / 32;                                     // K % 32 == 0 (checked on the host)
    const int per = (steps + GV_WAVES - 1) / GV_WAVES;
    const int s_beg = wave * per, s_end = min(steps, s_beg + per);
    const int n = n0 + idx;
    const bool n_ok = n < N, m_ok = idx < M;
    const bf16_t* wp = W + (long)(n_ok ? n : N - 1) * ldw + kg * 8;
    const bf16_t* ap = A + (long)(m_ok ? idx : 0) * lda + kg * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {};
    int s = s_beg;
    for (; s + GV_UNROLL <= s_end; s += GV_UNROLL) {
        bf16x8 wf[GV_UNROLL], af[GV_UNROLL];
#pragma unroll
        for (int u = 0; u < GV_UNROLL; ++u) wf[u] = *(const bf16x8*)(wp + (long)(s + u) * 32);
#pragma unroll
        for (int u = 0; u < GV_UNROLL; ++u) af[u] = m_ok ? *(const bf16x8*)(ap + (long)(s + u) * 32) : zero;
#pragma unroll
        for (int u = 0; u < GV_UNROLL; ++u)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(n_ok ? wf[u] : zero, af[u], acc, 0, 0, 0);   // D[n][m]
    }
    for (; s < s_end; ++s) {
        const bf16x8 wf = *(const bf16x8*)(wp + (long)s * 32);
        const bf16x8 af = m_ok ? *(const bf16x8*)(ap + (long)s * 32) : zero;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(n_ok ? wf : zero, af, acc, 0, 0, 0);
    }
    // D layout: lane holds D[row = kg*4 + r][col = idx] with rows = n (first operand), cols = m (second operand)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][kg * 4 + r][idx] = acc[r];
    __syncthreads();
    if (wave == 0) {
        // thread t < 256: (m = t / 16, nn = t % 16); 64 lanes x 4 passes
#pragma unroll
        for (int pss = 0; pss < 4; ++pss) {
            const int t = pss * 64 + lane, m = t >> 4, nn = t & 15;
            if (m < M && n0 + nn < N) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < GV_WAVES; ++w) v += part[w][nn][m];
                if (RESID) v = bf2f(R[(long)m * ldr + n0 + nn]) + rbf(v);     // torch: resid + bf16(x W^T)
                C[(long)m * ldc + n0 + nn] = f2bf(v);
            }
        }
    }
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
    const dim3 grid((N + 15) / 16), block(GV_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == 2)
        NV_LAUNCH(gemv_bf16_kernel<true>, grid, block, 0, st, (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, (const bf16_t*)R, M, N, K,
                  lda, ldw, ldc, ldr);
    else
        NV_LAUNCH(gemv_bf16_kernel<false>, grid, block, 0, st, (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, (const bf16_t*)nullptr, M, N,
                  K, lda, ldw, ldc, ldr);
    return nv_check_launch();
}

}  // extern "C"
