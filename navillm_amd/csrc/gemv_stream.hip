// Decode-step weight streamer (round 2): C[M <= 16, N] = pre(A)[M,K] @ W[N,K]^T (+R) with W in bf16 or weight-only fp8 (e4m3fn
// codes + per-row scales, semantics bf16(s*q) as everywhere in fp8w.hip).  One greedy-decoding step reads every weight of the LM
// once (13.5 GB at 7B bf16), so the kernel's job is to keep HBM streaming; tools/ubench/hbm_read.hip measures what the access
// shape allows on this part: 6.1-6.4 TB/s with whole 128-B lines per row, 5.4-5.6 TB/s with the 64-B half lines of the first
// GEMV (gemv_bf16.hip: 16 rows x 64 B per wave instruction = the MFMA operand layout straight from a row-major matrix).
//
//  * FULL LINES without a re-layout of W: a wave instruction loads 8 rows x 128 B.  Lane (i = lane&15, kg = lane>>4) holds row
//    i>>1, 16-B chunk c = (i&1)*4 + kg of the line.  As an MFMA operand that is "16 rows" where rows 2r and 2r+1 are the SAME
//    matrix row with different K chunks, so the x operand must differ between them:
//      M <= 8 : the x operand's 16 columns are (parity, m): column j = parity*8 + m carries x row m, chunks parity*4 + kg.
//               One MFMA per load serves both halves of the line; out[r][m] = D[2r][m] + D[2r+1][8+m].
//      M <= 16: two MFMAs per load with the x fragments of chunks 0-3 / 4-7 into two accumulators; out = D0[2r] + D1[2r+1].
//    The surplus MFMA work is free: the matrix pipe idles >95 % of a GEMV.
//  * ONE block per CU, each owning a contiguous run of 8-row units and the whole K: the x fragments of a K step are loaded once
//    per block and step (not once per 16 columns), every wave keeps two stages of <= 8 independent 1-KiB loads in flight
//    (stage g+2 is issued right after stage g is consumed, so the fp8 -> bf16 conversion overlaps the loads), and the
//    cross-wave reduction + store happens once per block.
//  * non-temporal loads for W (read once), so the x rows stay in L2.
//  * the decode layer's row kernels folded in (a layer is 6 launches instead of 11, navillm_amd/csrc/decoder_runtime.cpp):
//      - the x rows may be gathered (a_rows: the attention output still sitting in the cache frame);
//      - MODE >= 1: the operand is RMSNorm(A) = bf16(w * bf16(x * rstd)), rstd recomputed by every block with the arithmetic of
//        nv_rmsnorm_fwd_bf16 (same chunk order, same reduction tree: bit-identical operand), under the first W loads;
//      - MODE == 2: W = gate|up [2*ff, K]; a block owns gate unit p AND up unit ff/8 + p, and its epilogue writes
//        h = bf16(bf16(silu(g)) * u) of the bf16-rounded sums, as nv_swiglu_fwd_bf16 would from the stored gate|up -- which is
//        never written.  (Folding SwiGLU into the CONSUMER's operand instead was measured: every block re-evaluates silu for the
//        whole row, 256x redundant, and the down-projection GEMV went from 22.6 to 37.7 us.)
// Requirements of the fast path: K % 64 == 0 (bf16) / K % 128 == 0 (fp8), N % 8 == 0 (MODE 2: % 16), 16-B aligned rows; anything
// else is served by the generic kernels in gemv_bf16.hip / fp8w.hip (nvi_gemv_stream returns NV_ERR_SHAPE).
#include "nv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int GS_WAVES = 8;

__device__ __forceinline__ void fp8x4_f32(uint32_t w, float* f) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
    const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
__device__ __forceinline__ float gs_silu(float x) { return x / (1.f + expf(-x)); }    // == silu_f of lm_rowops.hip

// CFG: how a wave's loads are grouped into pipeline stages of <= 8 independent 1-KiB loads.  A stage = SG K-steps x UG units
// of one of the HG unit-halves.
template <int UMAX_> struct GsCfg;
template <> struct GsCfg<2>  { static constexpr int HG = 1, UG = 2, SG = 4; };
template <> struct GsCfg<4>  { static constexpr int HG = 1, UG = 4, SG = 2; };
template <> struct GsCfg<8>  { static constexpr int HG = 1, UG = 8, SG = 1; };
template <> struct GsCfg<16> { static constexpr int HG = 2, UG = 8, SG = 1; };

struct GsPre { const int* a_rows; const bf16_t* norm_w; float eps; };

template <bool FP8, bool WIDE, int UMAX, int MODE>
__global__ __launch_bounds__(GS_WAVES * 64) void gemv_stream_kernel(const bf16_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                                    const float* __restrict__ S, bf16_t* __restrict__ C,
                                                                    const bf16_t* __restrict__ R, int M, int N, int K, int lda,
                                                                    long ldw_bytes, int ldc, int ldr, int units_total, GsPre pre) {
    using Cfg = GsCfg<UMAX>;
    constexpr int HG = Cfg::HG, UG = Cfg::UG, SG = Cfg::SG, NST = 2;
    constexpr bool NORM = MODE >= 1, PAIR = MODE == 2;
    constexpr int KL = FP8 ? 128 : 64;                            // k values per 128-B line of W
    constexpr int NH = FP8 ? 2 : 1;                               // bf16x8 operand fragments per 16-B load
    constexpr int NP = WIDE ? 2 : 1;                              // accumulators (parities handled by separate MFMAs)
    constexpr int XF = NH * NP;                                   // x fragments per K-step
    extern __shared__ float part[];                               // [GS_WAVES][UMAX * 8][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 15, kg = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    // work items: 8-row units, or (PAIR) gate/up unit pairs -- item p = gate unit p + up unit units_total/2 + p
    const int items = PAIR ? units_total / 2 : units_total;
    const int i_beg = (int)((long)b * items / G), i_end = (int)((long)(b + 1) * items / G);
    const int nu = (i_end - i_beg) * (PAIR ? 2 : 1);              // units of this block, <= UMAX (host)
    auto unit_row = [&](int u) -> long {                          // first matrix row of the block's unit u
        return PAIR ? (long)(u & 1) * (N / 2) + (long)(i_beg + (u >> 1)) * 8 : (long)(i_beg + u) * 8;
    };
    const int steps = K / KL;
    const int per = (steps + GS_WAVES - 1) / GS_WAVES;
    const int s_beg = wave * per, s_end = min(steps, s_beg + per);
    const uint8_t* wp = W + (long)(idx >> 1) * ldw_bytes + ((idx & 1) * 4 + kg) * 16;
    const int xm = WIDE ? idx : (idx & 7), xpar = WIDE ? 0 : (idx >> 3);
    const bool m_ok = xm < M;
    // x fragment (p, h) of step s: k = s*KL + ((xpar + p)*4 + kg) * (KL/8) + h*8
    const int arow = m_ok ? (pre.a_rows ? pre.a_rows[xm] : xm) : 0;
    const int kl0 = (xpar * 4 + kg) * (KL / 8);                   // this lane's first k inside a line
    const bf16_t* ap = A + (long)arow * lda + kl0;
    const bf16_t* np = NORM ? pre.norm_w + kl0 : nullptr;
    float sc[UMAX];
    if (FP8) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) sc[u] = u < nu ? S[unit_row(u) + (idx >> 1)] : 0.f;
    }
    f32x4 acc[NP][UMAX];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < UMAX; ++u) acc[p][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero = {};
    float rstd = 0.f;
    struct Stage { u32x4 w[SG][UG]; bf16x8 x[SG][XF]; bf16x8 y[SG][NORM ? XF : 1]; };
    static_assert(NST % HG == 0, "a buffer always holds the same unit-half");
    Stage st[NST];
    const int nsteps = max(s_end - s_beg, 0);
    const int ngrp = ((nsteps + SG - 1) / SG) * HG;               // stages of this wave

    auto issue = [&](Stage& g, int gi, auto HC) {
        constexpr int h = decltype(HC)::value;
        if (gi >= ngrp) return;
        const int s0 = s_beg + (gi / HG) * SG;
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            if (s0 + j < s_end) {
#pragma unroll
                for (int u = 0; u < UG; ++u)
                    if (h * UG + u < nu)
                        g.w[j][u] = __builtin_nontemporal_load((const u32x4*)(wp + unit_row(h * UG + u) * ldw_bytes + (long)(s0 + j) * 128));
#pragma unroll
                for (int f = 0; f < XF; ++f) {
                    const int p = f / NH, hh = f % NH;
                    const long ko = (long)(s0 + j) * KL + p * 4 * (KL / 8) + hh * 8;
                    g.x[j][f] = m_ok ? *(const bf16x8*)(ap + ko) : zero;
                    if (NORM) g.y[j][f] = *(const bf16x8*)(np + ko);
                }
            }
        }
    };
    auto consume = [&](Stage& g, int gi, auto HC) {
        constexpr int h = decltype(HC)::value;
        if (gi >= ngrp) return;
        const int s0 = s_beg + (gi / HG) * SG;
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            if (s0 + j < s_end) {
                bf16x8 xo[XF];
#pragma unroll
                for (int f = 0; f < XF; ++f) {
                    if (!NORM) {
                        xo[f] = g.x[j][f];
                    } else {
                        const u32x4 xr = __builtin_bit_cast(u32x4, g.x[j][f]), yr = __builtin_bit_cast(u32x4, g.y[j][f]);
                        u32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x0 = __uint_as_float(xr[q] << 16), x1 = __uint_as_float(xr[q] & 0xffff0000u);
                            const float y0 = __uint_as_float(yr[q] << 16), y1 = __uint_as_float(yr[q] & 0xffff0000u);
                            o[q] = pack2bf(y0 * rbf(x0 * rstd), y1 * rbf(x1 * rstd));
                        }
                        xo[f] = __builtin_bit_cast(bf16x8, o);
                    }
                }
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    if (h * UG + u < nu) {
                        const int uu = h * UG + u;
                        bf16x8 wf[NH];
                        if (FP8) {
                            float f[16];
#pragma unroll
                            for (int q = 0; q < 4; ++q) fp8x4_f32(g.w[j][u][q], f + 4 * q);
                            u32x4 b0, b1;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                b0[q] = pack2bf(f[2 * q] * sc[uu], f[2 * q + 1] * sc[uu]);
                                b1[q] = pack2bf(f[8 + 2 * q] * sc[uu], f[8 + 2 * q + 1] * sc[uu]);
                            }
                            wf[0] = __builtin_bit_cast(bf16x8, b0);
                            wf[NH - 1] = __builtin_bit_cast(bf16x8, b1);
                        } else {
                            wf[0] = __builtin_bit_cast(bf16x8, g.w[j][u]);
                        }
#pragma unroll
                        for (int p = 0; p < NP; ++p)
#pragma unroll
                            for (int hh = 0; hh < NH; ++hh)
                                acc[p][uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[hh], xo[p * NH + hh], acc[p][uu], 0, 0, 0);   // D[i][col]
                    }
                }
            }
        }
    };
    // ring of NST stages in flight: buffer i always holds unit-half i % HG (compile-time accumulator indices)
#define GS_H(i) std::integral_constant<int, (i) % HG> {}
#define GS_STEP(i) consume(st[i], gi + (i), GS_H(i)); issue(st[i], gi + (i) + NST, GS_H(i));
    // RMSNorm operand: rstd of the M rows with the arithmetic of rmsnorm_fwd_kernel<4> -- 256 threads per row, thread t sums
    // chunks t*8 + i*2048 in order, wave tree, then the 4 wave sums left to right.  Two rows at a time (waves 0-3 / 4-7).  The x
    // loads go out BEFORE the first W stages (loads return in order: the reduction then waits for them only, with the W loads
    // still in flight) and the block-wide rendezvous happens while the memory pipe is full.
    constexpr int RMAX = WIDE ? 8 : 4;                            // rows per half-block
    constexpr int KCH = 4;                                        // 2048-element chunks per row: K <= 8192 (host)
    u32x4 nx[NORM ? RMAX : 1][NORM ? KCH : 1];
    if constexpr (NORM) {
        const int half = tid >> 8, t = tid & 255;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int m = 2 * r + half;
            const bf16_t* xr = A + (long)(m < M ? (pre.a_rows ? pre.a_rows[m] : m) : 0) * lda;
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                const int c = t * 8 + i * 2048;
                nx[r][i] = (m < M && c < K) ? *(const u32x4*)(xr + c) : u32x4{0, 0, 0, 0};
            }
        }
    }
    issue(st[0], 0, GS_H(0));
    issue(st[1], 1, GS_H(1));
    if constexpr (NORM) {
        __shared__ float red[16][4];
        const int half = tid >> 8;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int m = 2 * r + half;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = __uint_as_float(nx[r][i][q] << 16), b2 = __uint_as_float(nx[r][i][q] & 0xffff0000u);
                    ss += a * a;
                    ss += b2 * b2;
                }
            }
            if (m < M) {
                const float w = wave_sum(ss);
                if (lane == 0) red[m][wave & 3] = w;
            }
        }
        __syncthreads();
        if (m_ok) {
            float tt = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) tt += red[xm][i];
            rstd = rsqrtf(tt / (float)K + pre.eps);
        }
    }
    for (int gi = 0; gi < ngrp; gi += NST) {
        GS_STEP(0)
        GS_STEP(1)
    }
#undef GS_STEP
#undef GS_H
    // D layout: lane (idx = column, kg) holds D[i = kg*4 + e][column]; row i = 2r + parity(i).
    //   narrow: column = parity_x*8 + m : lanes idx < 8 own the even rows (e = 0, 2), lanes idx >= 8 the odd rows (e = 1, 3)
    //   wide  : acc[0] even rows, acc[1] odd rows, summed in-lane
    float (*pw)[17] = (float (*)[17])(part + (size_t)wave * UMAX * 8 * 17);
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        if (u < nu) {
            if (WIDE) {
                pw[u * 8 + kg * 2 + 0][idx] = acc[0][u][0] + acc[NP - 1][u][1];
                pw[u * 8 + kg * 2 + 1][idx] = acc[0][u][2] + acc[NP - 1][u][3];
            } else {
                pw[u * 8 + kg * 2 + 0][idx] = idx < 8 ? acc[0][u][0] : acc[0][u][1];
                pw[u * 8 + kg * 2 + 1][idx] = idx < 8 ? acc[0][u][2] : acc[0][u][3];
            }
        }
    }
    __syncthreads();
    auto total = [&](int c, int m) {                              // sum over the 8 K-slices of block column c, x row m
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < GS_WAVES; ++wv) {
            const float* q = part + ((size_t)wv * UMAX * 8 + c) * 17;
            v += WIDE ? q[m] : q[m] + q[m + 8];
        }
        return v;
    };
    if (PAIR) {
        const int ncols = (nu / 2) * 8;                           // h columns of this block
        const long n0 = (long)i_beg * 8;
        for (int e = tid; e < 16 * ncols; e += GS_WAVES * 64) {
            const int m = e / ncols, c = e - m * ncols;
            if (m < M) {
                const int q = c >> 3, r = c & 7;
                const float gate = rbf(total((2 * q) * 8 + r, m)), up = rbf(total((2 * q + 1) * 8 + r, m));
                C[(long)m * ldc + n0 + c] = f2bf(rbf(gs_silu(gate)) * up);
            }
        }
    } else {
        const int ncols = nu * 8;
        const long n0 = (long)i_beg * 8;
        for (int e = tid; e < 16 * ncols; e += GS_WAVES * 64) {
            const int m = e / ncols, c = e - m * ncols;
            if (m < M) {
                float v = total(c, m);
                if (R) v = bf2f(R[(long)m * ldr + n0 + c]) + rbf(v);      // torch: resid + bf16(x W^T)
                C[(long)m * ldc + n0 + c] = f2bf(v);
            }
        }
    }
}

int num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <bool FP8, bool WIDE, int UMAX, int MODE>
int launch(const void* A, const void* W, const float* S, void* C, const void* R, int M, int N, int K, int lda, long ldw_bytes, int ldc,
           int ldr, int units, int grid, const GsPre& pre, hipStream_t st) {
    auto kern = gemv_stream_kernel<FP8, WIDE, UMAX, MODE>;
    constexpr int LDS = GS_WAVES * UMAX * 8 * 17 * 4;
    static int attr = -1;                                          // per instantiation
    if (LDS > 48 * 1024 && attr != 0) {
        attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess ? 0 : 1;
        if (attr != 0) return NV_ERR_LAUNCH;
    }
    NV_LAUNCH(kern, dim3(grid), dim3(GS_WAVES * 64), LDS, st, (const bf16_t*)A, (const uint8_t*)W, S, (bf16_t*)C, (const bf16_t*)R, M, N, K,
              lda, ldw_bytes, ldc, ldr, units, pre);
    return nv_check_launch();
}

template <bool FP8, bool WIDE, int MODE>
int dispatch(const void* A, const void* W, const float* S, void* C, const void* R, int M, int N, int K, int lda, long ldw_bytes, int ldc,
             int ldr, const GsPre& pre, hipStream_t st) {
    constexpr int UCAP = WIDE ? 8 : 16;                            // two accumulators per unit: half as many units per block
    constexpr int PER = MODE == 2 ? 2 : 1;                         // units per work item (gate + up)
    const int units = N / 8, items = units / PER;
    int grid = items < num_cus() ? items : num_cus();
    int umax = (items + grid - 1) / grid * PER;
    if (umax > UCAP) { grid = (items + UCAP / PER - 1) / (UCAP / PER); umax = UCAP; }
    // (two stages in flight per wave; four measured no better: tools/gemv_fp8_probe.py)
#define NV_GS(U) return launch<FP8, WIDE, U, MODE>(A, W, S, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, units, grid, pre, st)
    if (umax <= 2) NV_GS(2);
    if (umax <= 4) NV_GS(4);
    if (umax <= 8 || WIDE) NV_GS(8);
    NV_GS((WIDE ? 8 : 16));
#undef NV_GS
}

template <int MODE>
int dispatch_mode(const void* A, const void* W, const float* S, void* C, const void* R, int M, int N, int K, int lda, long ldw_bytes, int ldc,
                  int ldr, int fp8, const GsPre& pre, hipStream_t st) {
    if (fp8) return M > 8 ? dispatch<true, true, MODE>(A, W, S, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, pre, st)
                          : dispatch<true, false, MODE>(A, W, S, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, pre, st);
    return M > 8 ? dispatch<false, true, MODE>(A, W, S, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, pre, st)
                 : dispatch<false, false, MODE>(A, W, S, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, pre, st);
}

}  // namespace

extern "C" int nvi_gemv_stream(const void* A, const void* W, const float* scales, void* C, const void* R, int M, int N, int K, int lda,
                               int ldw, int ldc, int ldr, int resid, int fp8, void* stream) {
    static const bool off = [] { const char* e = getenv("NV_GEMV_STREAM"); return e && atoi(e) == 0; }();   // A/B knob
    if (off) return NV_ERR_SHAPE;
    if (resid && !R) return NV_ERR_ARG;
    return nv_gemv_pre(A, nullptr, W, fp8 ? scales : nullptr, C, resid ? R : nullptr, M, N, K, lda, ldw, ldc, ldr, 0, nullptr, 0.f, 0, stream);
}

// C = pre(A)[M <= 16, K] @ W[N, K]^T: the decode-step Linear with its row kernels folded in (modes above).
//   W bf16 [N, ldw] (scales == NULL) or e4m3fn codes [N, ldw] + per-row scales;  a_rows: optional gather of A's rows;
//   rmsnorm != 0: the operand is RMSNorm(A; norm_w [K] bf16, eps);
//   swiglu == 0: C [M, N] (+ R [M, ldr] residual, optional);  swiglu != 0 (needs rmsnorm, no R): W = gate|up, C [M, N/2] = SwiGLU.
// NV_ERR_SHAPE when the shape is outside the streamer's fast path (K % 64 / % 128 for fp8, N % 8, 16-B aligned rows, M <= 16).
extern "C" int nv_gemv_pre(const void* A, const int* a_rows, const void* W, const float* scales, void* C, const void* R, int M, int N, int K,
                           int lda, int ldw, int ldc, int ldr, int rmsnorm, const void* norm_w, float eps, int swiglu, void* stream) {
    if (!A || !W || !C || (rmsnorm && !norm_w) || (swiglu && (!rmsnorm || R))) return NV_ERR_ARG;
    if (rmsnorm && K > 8192) return NV_ERR_SHAPE;
    const int fp8 = scales != nullptr;
    if (M < 1 || M > 16 || N < 8 || (N & 7) || (lda & 7) || (swiglu && (N & 15))) return NV_ERR_SHAPE;
    if (fp8 ? ((K & 127) || (ldw & 15)) : ((K & 63) || (ldw & 7))) return NV_ERR_SHAPE;
    if ((((uintptr_t)A) | ((uintptr_t)W) | ((uintptr_t)norm_w)) & 15) return NV_ERR_SHAPE;
    const long ldw_bytes = fp8 ? (long)ldw : (long)ldw * 2;
    hipStream_t st = (hipStream_t)stream;
    const GsPre gp{a_rows, (const bf16_t*)norm_w, eps};
    if (swiglu) return dispatch_mode<2>(A, W, scales, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, fp8, gp, st);
    if (rmsnorm) return dispatch_mode<1>(A, W, scales, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, fp8, gp, st);
    return dispatch_mode<0>(A, W, scales, C, R, M, N, K, lda, ldw_bytes, ldc, ldr, fp8, gp, st);
}
