// bf16 MFMA GEMM family for the visual-token causal LM (K7a in SURVEY.md §2.2).
//
//   C[M,N] = sum_k Aop[m,k] * Bop[n,k]          fp32 accumulate, bf16 in/out
//
// Each operand is either K-major ("KMAJ": stored [rows][K], the contraction index is
// contiguous -- nn.Linear's x and W) or MN-major (stored [K][rows], the free index is
// contiguous).  That one switch per operand gives the three GEMMs the LM needs without
// ever materialising a transpose in HBM:
//     forward   y  = x  @ W^T   : A=x  KMAJ,  B=W  KMAJ     (reference: HF LlamaAttention/MLP
//     dgrad     dx = dy @ W     : A=dy KMAJ,  B=W  MN-major   Linear calls reached from
//     wgrad     dW = dy^T @ x   : A=dy MN,    B=x  MN-major   models/modified_lm.py:112-116)
//
// CDNA4 design: 64-lane waves, v_mfma_f32_16x16x32_bf16, BK=64 K-tiles streamed HBM->LDS by
// bounds-checked `buffer_load_dwordx4 ... lds` (no VGPR round trip, OOB rows read as 0 so
// ragged M/N need no padding), double-buffered LDS, XOR-swizzled LDS images (swizzle applied on
// the per-lane SOURCE address, LDS writes stay lane-linear), KMAJ fragments by ds_read_b128,
// MN-major fragments by the gfx950 transposing read ds_read_b64_tr_b16, XCD-aware tile order.
// MFMA operands are swapped (D[n][m]) so each lane ends with 4 consecutive n: 8-byte stores
// and room for fused row-wise epilogues.
#include "nv_common.h"

namespace {

constexpr int BK = 64;

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_RESID = 2, EPI_BIAS = 3 };

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const bf16_t* R;        // EPI_RESID: residual [M,N] (ldr) ; EPI_BIAS: bias[N]
    int M, N, K;
    int lda, ldb, ldc, ldr;
    uint32_t a_bytes, b_bytes;
};

// ---- LDS images -------------------------------------------------------------------------
// KMAJ tile  : [R][64] bf16, 128-B rows = 8 slots of 16 B; phys_slot = slot ^ ((row>>1)&7)
//              -> a ds_read_b128 lane group (16 distinct rows, 2 adjacent slots) is conflict-free.
// MN tile    : [64][R] bf16, 2R-B rows; phys_slot = slot ^ (key(krow)<<1),
//              key = (krow&3) | ((krow>>3)&1)<<2 -> the 8 k-rows one half-wave touches in a
//              ds_read_b64_tr_b16 land on 8 distinct 32-B bank groups.
__device__ __forceinline__ int mn_key(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// Issue the loads of one [R x 64] operand tile into LDS (all NT threads cooperate).
template <int R, bool KMAJ, int NT>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, LDS_PTR(char) lds, int row0, int k0,
                                           int ld, int tid) {
    constexpr int TILE_BYTES = R * BK * 2;
    constexpr int ITERS = TILE_BYTES / (NT * 16);
    static_assert(TILE_BYTES % (NT * 16) == 0, "tile/threads mismatch");
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        // this wave-instruction fills LDS bytes [chunk*1024, chunk*1024+1024)
        const int chunk = it * (NT / 64) + wave;
        uint32_t voff;
        if (KMAJ) {
            const int row = chunk * 8 + (lane >> 3);           // 8 rows of 128 B per KiB
            const int slot = (lane & 7) ^ ((row >> 1) & 7);    // logical slot living at phys slot lane&7
            voff = (uint32_t)(((long)(row0 + row) * ld + k0 + slot * 8) * 2);
        } else {
            constexpr int SLOTS = R / 8;                       // 16-B slots per k-row
            constexpr int ROWS_PER_KIB = 64 / SLOTS;
            const int krow = chunk * ROWS_PER_KIB + lane / SLOTS;
            const int slot = (lane % SLOTS) ^ (mn_key(krow) << 1);
            voff = (uint32_t)(((long)(k0 + krow) * ld + row0 + slot * 8) * 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))(lds + chunk * 1024), 16, voff, 0, 0, 0);
    }
}

// One MFMA operand fragment (16 rows x 32 k) from an LDS tile.
template <int R, bool KMAJ>
__device__ __forceinline__ bf16x8 load_frag(LDS_PTR(char) tile, int r0, int kk, int lane) {
    const int idx = lane & 15, kg = lane >> 4;
    if (KMAJ) {
        const int row = r0 + idx;
        const int slot = (kk * 4 + kg) ^ ((row >> 1) & 7);
        return *(LDS_PTR(bf16x8))(tile + row * 128 + slot * 16);
    } else {
        // transposing read: lane t of a 16-lane group hands in the address of row (t>>2),
        // 8-byte chunk (t&3) of a [4 k][16 col] block and receives column t (4 k values).
        const int kr = kk * 32 + kg * 8 + (idx >> 2);
        const int slot = (r0 >> 3) + ((idx & 3) >> 1);
        const int half = (idx & 1) * 8;
        const int a0 = kr * (R * 2) + ((slot ^ (mn_key(kr) << 1)) << 4) + half;
        const int kr2 = kr + 4;
        const int a1 = kr2 * (R * 2) + ((slot ^ (mn_key(kr2) << 1)) << 4) + half;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + a1));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int BM, int BN, int WGM, int WGN, bool A_KMAJ, bool B_KMAJ, int EPI>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order; N-tiles fastest so consecutive ids share the A row panel in L2.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = t / tiles_n, tn = t % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = (p.K + BK - 1) / BK;
    auto stage = [&](int kt, int buf) {
        LDS_PTR(char) sa = smem + buf * (A_BYTES + B_BYTES);
        stage_tile<BM, A_KMAJ, NT>(ra, sa, m0, kt * BK, p.lda, tid);
        stage_tile<BN, B_KMAJ, NT>(rb, sa + A_BYTES, n0, kt * BK, p.ldb, tid);
    };
    auto compute = [&](int buf) {
        LDS_PTR(char) sa = smem + buf * (A_BYTES + B_BYTES);
        LDS_PTR(char) sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[j] = load_frag<BM, A_KMAJ>(sa, wm * WTM + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = load_frag<BN, B_KMAJ>(sb, wn * WTN + i * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
        }
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT - 1; ++kt) {
        stage(kt + 1, cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue: lane holds D[n = g*4+r][m = lane&15] per 16x16 tile ----
    const int g = lane >> 4, mi = lane & 15;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 16 + mi;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WTN + i * 16 + g * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            bf16_t* cp = p.C + (long)m * p.ldc + n;
            const bool full = (n + 3 < p.N);
            if (EPI == EPI_BIAS) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] += bf2f(p.R[n + r]);
            } else if (EPI == EPI_ACCUM) {
                // torch semantics of `grad += dW`: dW is rounded to bf16 first, then added
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] = bf2f(cp[r]) + rbf(v[r]);
            } else if (EPI == EPI_RESID) {
                const bf16_t* rp = p.R + (long)m * p.ldr + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] = bf2f(rp[r]) + rbf(v[r]);
            }
            if (full && ((((uintptr_t)cp) & 7) == 0)) {
                u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                *(u32x2*)cp = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) cp[r] = f2bf(v[r]);
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, bool A_KMAJ, bool B_KMAJ, int EPI>
int launch(const GemmArgs& p, hipStream_t st) {
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_bf16_kernel<BM, BN, WGM, WGN, A_KMAJ, B_KMAJ, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return NV_ERR_LAUNCH;
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    NV_LAUNCH(kern, dim3(tiles), dim3(WGM * WGN * 64), LDS, st, p);
    return nv_check_launch();
}

template <bool A_KMAJ, bool B_KMAJ, int EPI>
int dispatch_tile(const GemmArgs& p, int tile_cfg, hipStream_t st) {
    // tile_cfg: 0 = auto, 1 = 128x128 (4 waves), 2 = 256x128 (8 waves), 3 = 256x256 (8 waves)
    if (tile_cfg == 0) {
        // Fill the chip first: 256 CUs; the big tile only when it still gives >= ~2 rounds.
        // measured on MI355X (profiles/r01_gemm_probe.txt): the 256x256 tile wins from ~1.4 rounds
        // of the 256 CUs upward, in all three layouts
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        tile_cfg = (t256 >= 128) ? 3 : 1;
    }
    switch (tile_cfg) {
        case 1: return launch<128, 128, 2, 2, A_KMAJ, B_KMAJ, EPI>(p, st);
        case 2: return launch<256, 128, 4, 2, A_KMAJ, B_KMAJ, EPI>(p, st);
        case 3: return launch<256, 256, 2, 4, A_KMAJ, B_KMAJ, EPI>(p, st);
    }
    return NV_ERR_ARG;
}

template <bool A_KMAJ, bool B_KMAJ>
int dispatch_epi(const GemmArgs& p, int epi, int tile_cfg, hipStream_t st) {
    switch (epi) {
        case EPI_STORE: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_STORE>(p, tile_cfg, st);
        case EPI_ACCUM: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_ACCUM>(p, tile_cfg, st);
        case EPI_RESID: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_RESID>(p, tile_cfg, st);
        case EPI_BIAS: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_BIAS>(p, tile_cfg, st);
    }
    return NV_ERR_ARG;
}

inline uint32_t span_bytes(long rows, long cols, long ld) {
    if (rows <= 0) return 0;
    // bounds are checked per dword: round the last row up to an even element count (ld % 8 == 0
    // guarantees that extra element still lies inside the row)
    const long b = ((rows - 1) * ld + ((cols + 1) & ~1L)) * 2;
    return b > 0xffffffffL ? 0xffffffffu : (uint32_t)b;
}

}  // namespace

// ---------------------------------------------------------------- C ABI (see include/navillm_hip.h)
extern "C" int nv_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                            int lda, int ldb, int ldc, int ldr, int epilogue, int tile_cfg, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    if ((epilogue == EPI_RESID || epilogue == EPI_BIAS) && !R) return NV_ERR_ARG;
    if ((lda & 7) || (ldb & 7)) return NV_ERR_SHAPE;                 // 16-B aligned rows for the DMA
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return NV_ERR_SHAPE;
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.R = (const bf16_t*)R;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    hipStream_t st = (hipStream_t)stream;
    switch (layout) {
        case 0:  // NT: A[M,K], B[N,K]
            if (K % BK) return NV_ERR_SHAPE;
            p.a_bytes = span_bytes(M, K, lda); p.b_bytes = span_bytes(N, K, ldb);
            return dispatch_epi<true, true>(p, epilogue, tile_cfg, st);
        case 1:  // NN: A[M,K], B[K,N]
            if (K % BK) return NV_ERR_SHAPE;
            p.a_bytes = span_bytes(M, K, lda); p.b_bytes = span_bytes(K, N, ldb);
            return dispatch_epi<true, false>(p, epilogue, tile_cfg, st);
        case 2:  // TN: A[K,M], B[K,N]   (K = contraction, any size: OOB k-rows read as zero)
            p.a_bytes = span_bytes(K, M, lda); p.b_bytes = span_bytes(K, N, ldb);
            return dispatch_epi<false, false>(p, epilogue, tile_cfg, st);
    }
    return NV_ERR_ARG;
}
