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
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include <stdio.h>

namespace {

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_RESID = 2, EPI_BIAS = 3, EPI_SWIGLU_BWD = 4, EPI_ROPE = 5 };

constexpr int MAX_SLABS = 512;   // fp32 partial tiles the split-K tail may have in flight (workspace = MAX_SLABS * 256 KiB + 4 KiB)

// C-tile store: 0 plain, 1 non-temporal (default: C is consumed by the NEXT kernel, whose blocks mostly sit on other XCDs), 2
// write-through (sc1) -- NV_GEMM_C_STORE.  Measured on the training step: nt +0.3 %, sc1 +-0 (the idea that the ~6 us idle gap
// after every kernel is the write-back of dirty C lines did not hold: write-through C leaves the gaps where they were).
__device__ __forceinline__ void store_c(bf16_t* cp, const u32x4& t, int mode) {
    if (mode == 1) __builtin_nontemporal_store(t, (u32x4*)cp);
    else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cp), "v"(t) : "memory");
    else *(u32x4*)cp = t;
}

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const bf16_t* R;        // EPI_RESID: residual [M,N] (ldr) ; EPI_BIAS: bias[N] ; EPI_SWIGLU_BWD: gate|up [M,2N] (ldr)
    int M, N, K;
    int lda, ldb, ldc, ldr;
    uint32_t a_bytes, b_bytes;
    int group_m;            // tile-order super-row height (L2 reuse), >= 1
    int col_strips;         // 1: walk column strips of 8 tiles (XCDs partition B), 0: row groups (XCDs partition A)
    // split-K tail (see launch()): blocks >= full_blocks are K-slices of the last, partial round of tiles
    int full_blocks, rem, split;
    int c_nt;               // C-tile store mode (store_c)
    int band_reduce;        // split-K hand-off: 1 = every slice reduces one row band of its tile (default), 0 = the last arriver reduces all
    float* slabs;           // [rem*split][BM*BN] fp32 partials
    unsigned* counters;     // [2][512] arrival / departure tickets per tail tile, zero between launches
    int debug;              // NV_GEMM_DEBUG (measurement only): bit0 skip the C stores, bit1 skip the K loop, bit2 record clocks
    int items, persist;     // work items of the launch (tiles + tail K-slices); persistent-block mode on/off
    // EPI_ROPE (packed q|k|v projection): rotate the q and k columns (n < rope_cols) as they are written; position of row
    // m is m % rope_S; R = cos table, rope_sin = sin table, both [maxS][128] bf16 as nv_rope_bf16 takes them
    const bf16_t* rope_sin; int rope_S, rope_cols;
    const int* rope_pos;    // optional: position of row m (packed rows); nullptr -> m % rope_S
    // weight-only fp8 B operand (PIPE 7 / 8, nv_gemm_fp8w): B = e4m3fn codes [N][K] (ldb in bytes), one fp32 scale per output channel n
    const float* b_scales;
    int epi_preload;        // 1 (default): the reading epilogues issue their global loads eight passes ahead (NV_GEMM_EPI_PRELOAD=0: one pass at a time)
    int fp8_epi;            // PIPE 8 only: 1 = the codes are converted unscaled and s[n] multiplies the fp32 accumulator in the epilogue
};

// ---- LDS images (BKT = K extent of a stage, 64 or 32) -----------------------------------
// KMAJ tile  : [R][BKT] bf16; 16-B slots per row S = BKT/8; phys_slot = slot ^ swz(row) with
//              swz = (row>>1)&7 (BKT=64) or (row>>2)&3 (BKT=32): the 16 rows a ds_read_b128 lane
//              group touches land on 16 distinct 16-B bank slots -> conflict-free (PMC: 0).
// MN tile    : [BKT][R] bf16, 2R-B rows; phys_slot = slot ^ (key(krow)<<1),
//              key = (krow&3) | ((krow>>3)&1)<<2 -> the 8 k-rows one half-wave touches in a
//              ds_read_b64_tr_b16 land on 8 distinct 32-B bank groups.
__device__ __forceinline__ int mn_key(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
template <int BKT>
__device__ __forceinline__ int km_swz(int row) { return BKT == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// Source byte offset (from the operand base) of the 16 B this lane moves in wave-instruction `it` of an operand
// tile's DMA; that instruction fills LDS bytes [chunk*1024, +1024) of the [R x BKT] image, chunk = it*(NT/64)+wave.
template <int R, bool KMAJ, int NT, int BKT>
__device__ __forceinline__ uint32_t piece_voff(int row0, int k0, int ld, int tid, int it) {
    const int wave = tid >> 6, lane = tid & 63;
    const int chunk = it * (NT / 64) + wave;
    if (KMAJ) {
        constexpr int SLOTS = BKT / 8;                     // 16-B slots per row
        constexpr int RPK = 64 / SLOTS;                    // rows per KiB
        const int row = chunk * RPK + lane / SLOTS;
        const int slot = (lane % SLOTS) ^ km_swz<BKT>(row);  // logical slot living at this phys slot
        return (uint32_t)(((long)(row0 + row) * ld + k0 + slot * 8) * 2);
    } else {
        constexpr int SLOTS = R / 8;                       // 16-B slots per k-row
        constexpr int ROWS_PER_KIB = 64 / SLOTS;
        const int krow = chunk * ROWS_PER_KIB + lane / SLOTS;
        const int slot = (lane % SLOTS) ^ (mn_key(krow) << 1);
        return (uint32_t)(((long)(k0 + krow) * ld + row0 + slot * 8) * 2);
    }
}
template <int R, bool KMAJ, int NT, int BKT>
__device__ __forceinline__ void stage_piece(const u32x4& desc, uint32_t lds, int row0, int k0, int ld, int tid, int it) {
    const int chunk = it * (NT / 64) + (tid >> 6);
    dma16_nosave(desc, __builtin_amdgcn_readfirstlane(lds + chunk * 1024), piece_voff<R, KMAJ, NT, BKT>(row0, k0, ld, tid, it));
}

// Issue the loads of one [R x BKT] operand tile into LDS (all NT threads cooperate).
template <int R, bool KMAJ, int NT, int BKT>
__device__ __forceinline__ void stage_tile(const u32x4& desc, uint32_t lds, int row0, int k0, int ld, int tid) {
    constexpr int TILE_BYTES = R * BKT * 2;
    constexpr int ITERS = TILE_BYTES / (NT * 16);
    static_assert(TILE_BYTES % (NT * 16) == 0, "tile/threads mismatch");
#pragma unroll
    for (int it = 0; it < ITERS; ++it) stage_piece<R, KMAJ, NT, BKT>(desc, lds, row0, k0, ld, tid, it);
}

// One MFMA operand fragment (16 rows x 32 k) from an LDS tile.
template <int R, bool KMAJ, int BKT>
__device__ __forceinline__ bf16x8 load_frag(LDS_PTR(char) tile, int r0, int kk, int lane) {
    const int idx = lane & 15, kg = lane >> 4;
    if (KMAJ) {
        const int row = r0 + idx;
        const int slot = (kk * 4 + kg) ^ km_swz<BKT>(row);
        return *(LDS_PTR(bf16x8))(tile + row * (BKT * 2) + slot * 16);
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

// wait until at most `tiles` whole stages (LOADS buffer_loads each) are still in flight
// Per-lane fragment addressing for the software-pipelined main loop: everything that depends on the lane is
// computed ONCE (2 VGPRs for a K-major operand, one per fragment for an MN-major one); tile index, k-step
// and the second transposing read are immediates on the ds_read.
// FS = distance, in 16-row fragments, between a wave's consecutive fragments (1: the wave owns a contiguous block of rows;
// WGM: the waves' fragment rows are interleaved, fragment j of wave-row wm is tile fragment j*WGM + wm -- see gemm_tile).
template <int R, bool KMAJ, int NF, int FS = 1>
struct FragAddr {
    int off[KMAJ ? 2 : NF];
    __device__ __forceinline__ void init(int w0, int lane) {     // w0 = wave's first row/col in the tile (% 16 == 0)
        const int idx = lane & 15, kg = lane >> 4;
        if (KMAJ) {
            const int s_ = (idx >> 1) & 7;                        // km_swz<64>(row): depends on idx only
            off[0] = (w0 + idx) * 128 + ((kg ^ s_) << 4);
            off[1] = off[0] ^ 64;                                 // slot (4+kg)^s
        } else {
            const int key = ((idx >> 2) & 3) | ((kg & 1) << 2);   // mn_key(krow): the same for both reads and both k-steps
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int slot = ((w0 >> 3) + 2 * FS * j + ((idx & 3) >> 1)) ^ (key << 1);
                off[KMAJ ? 0 : j] = (kg * 8 + (idx >> 2)) * (R * 2) + (slot << 4) + (idx & 1) * 8;
            }
        }
    }
    __device__ __forceinline__ bf16x8 load(LDS_PTR(char) tile, int j, int kk) const {
        if (KMAJ) {
            return *(LDS_PTR(bf16x8))(tile + off[KMAJ ? kk : 0] + j * (2048 * FS));
        } else {
            LDS_PTR(char) q = tile + off[KMAJ ? 0 : j] + kk * (32 * R * 2);
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))q);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(q + 4 * R * 2));
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        }
    }
};

// The same for the interleaved loop, with the LDS stage folded into the per-lane address: `off` addresses the stage
// being READ NEXT for k-step 0 / CURRENTLY for k-step 1 (see the loop), and flip() toggles it with one v_xor per
// address register per K-tile -- all the vector ALU work that is left in that loop.  Every read is then
// `ds_read vdst, vaddr offset:imm`.  asm("" : "+v") pins each address in its own VGPR (hipcc otherwise re-derives
// them with v_add chains inside the loop).
template <int R, bool KMAJ, int NF, int STAGE_STRIDE, int FS = 1>
struct FragAddr2 {
    static_assert((STAGE_STRIDE & (STAGE_STRIDE - 1)) == 0, "stage stride must be a power of two (xor toggle)");
    uint32_t off[KMAJ ? 2 : NF];
    __device__ __forceinline__ void init(int w0, int lane, uint32_t region) {   // region: LDS byte address of this operand's stage 0
        FragAddr<R, KMAJ, NF, FS> f;
        f.init(w0, lane);
#pragma unroll
        for (int q = 0; q < (KMAJ ? 2 : NF); ++q) {
            off[q] = region + f.off[q];
            asm volatile("" : "+v"(off[q]));
        }
    }
    __device__ __forceinline__ void flip() {
#pragma unroll
        for (int q = 0; q < (KMAJ ? 2 : NF); ++q) {
            off[q] ^= STAGE_STRIDE;
            asm volatile("" : "+v"(off[q]));
        }
    }
    __device__ __forceinline__ bf16x8 load(int j, int kk) const {
        if (KMAJ) {
            return *(LDS_PTR(bf16x8))(uintptr_t)(off[KMAJ ? kk : 0] + j * (2048 * FS));
        } else {
            const uint32_t q = off[KMAJ ? 0 : j] + kk * (32 * R * 2);
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)q);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)(q + 4 * R * 2));
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        }
    }
};

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int LOADS, int MAXT>
__device__ __forceinline__ void wait_tiles(int tiles) {
    if (MAXT >= 3 && tiles >= 3) wait_vmcnt<3 * LOADS>();
    else if (MAXT >= 2 && tiles == 2) wait_vmcnt<2 * LOADS>();
    else if (MAXT >= 1 && tiles == 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
}

// SwiGLU backward on one element (the arithmetic and rounding points of swiglu_bwd_kernel, lm_rowops.hip):
// dh arrives rounded to bf16, like the materialised dh tensor it replaces.
__device__ __forceinline__ void swiglu_bwd_elem(float dh, float g, float u, float& dg, float& du) {
    const float sg = 1.f / (1.f + expf(-g));
    du = dh * rbf(g * sg);
    dg = rbf(dh * u) * (sg * (1.f + g * (1.f - sg)));
}

// One work item (an output tile, or one K-slice of a tail tile) of the GEMM.  `first` = this block's first item: later
// items of a persistent block start with the previous tile's C stores still in flight (see the kernel below).
// TME (interleaved loop, PIPE 4, only): fragment rows per wave actually computed.  There the waves' 16-row fragments are
// INTERLEAVED along M -- fragment j of wave-row wm is tile fragment j*WGM + wm -- so a tile may be cut off after any even number
// of fragments and both wave-rows still carry the same load: BM_EFF = WGM*TME*16 rows (256, 224, 192, 160, 128 for TME = 8..4)
// at TME/8 of the MFMA work.  The launcher picks TME per GEMM shape so that the tile count fills the 256 CUs' rounds
// (small-M GEMMs: M = 670 -> 5 tile rows of 160 instead of 3 of 256, of which the last is 62 % padding; see plan_tme()).
template <int BM, int BN, int WGM, int WGN, int BKT, int NSTAGE, bool A_KMAJ, bool B_KMAJ, int EPI, int PIPE, int TME>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, const int item, const bool first, LDS_PTR(char) smem) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr bool IL = (PIPE == 4 || PIPE >= 6);          // interleaved fragment rows (hand-scheduled loops)
    constexpr bool BF8 = PIPE >= 7;                        // B = weight-only fp8 codes + per-row scales (three-stage loop of the cut-off tiles)
    constexpr int TMU = IL ? TME : TM;                     // fragment rows per wave in use
    constexpr int BM_EFF = IL ? WGM * TME * 16 : BM;       // rows of C this tile covers
    static_assert(TME >= 1 && TME <= TM && (IL || TME == TM), "TME < TM needs the interleaved loop");
    constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2;
    constexpr int LOADS = (A_BYTES + B_BYTES) / (NT * 16);     // buffer_load...lds per thread per stage
    static_assert(3 * LOADS < 64, "vmcnt immediate range");

    // opaque per tile: otherwise hipcc hoists every lane-dependent address of the prologue/epilogue out of the persistent
    // tile loop, cannot keep them next to the 128 accumulators and spills them (+6 % kernel time measured)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // Tile order: each XCD (block b runs on XCD b%8) gets a contiguous chunk of a GROUPED order in which
    // consecutive ids walk group_m tile-rows down, then one tile-column right.  The ~32 tiles an XCD has
    // in flight then form a group_m x (32/group_m) patch sharing A row panels AND B column panels in
    // that XCD's 4 MiB L2 (measured: strip order = 49% L2 hit rate on the forward GEMM).
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM_EFF - 1) / BM_EFF;
    int tile_b = item, ks = 0, nsplit = 1, tail_u = 0;
    if (tile_b >= p.full_blocks) {
        // K-slices of the last, partial round.  Tail item q -> XCD x = q % 8 (block b runs on XCD b % 8; full_blocks % 8 == 0 is not
        // needed for correctness) and position idx = q / 8 inside that XCD's run; the run holds the XCD's contiguous chunk of tail
        // tiles, ALL slices of a tile next to each other (idx = tile * split + slice).  So the slices of a tile are dispatched
        // within 8 * split consecutive blocks -- the band reduction below makes them wait for one another, and at most one tile per
        // XCD can be partly dispatched at any time -- and the same-slice blocks an XCD runs side by side work on neighbouring tiles
        // at the same K range (shared operand panels in that XCD's L2).
        const int q = tile_b - p.full_blocks;
        const int x = q & 7, idx = q >> 3;                             // full_blocks % 256 == 0: x IS this block's XCD
        const int per = p.rem >> 3, extra = p.rem & 7;                 // tail tiles of XCD x: per (+1 for x < extra)
        const int mine_tiles = per + (x < extra ? 1 : 0);
        if (idx >= mine_tiles * p.split) return;                       // padding item (the tail is padded to whole groups of 8)
        ks = idx % p.split;
        tail_u = 8 * (idx / p.split) + x;                              // the tile whose un-split block id would ALSO land on XCD x: xcd_remap
                                                                       // below then continues that XCD's contiguous chunk of tiles
        tile_b = p.full_blocks + tail_u;
        nsplit = p.split;
    }
    // block-uniform by construction; say so (the integer divisions above run on the vector ALU, and with the bounded spin + trap
    // further down hipcc otherwise loses track and hands VGPRs to the "s" operands of the DMA asm)
    ks = __builtin_amdgcn_readfirstlane(ks);
    tail_u = __builtin_amdgcn_readfirstlane(tail_u);
    tile_b = __builtin_amdgcn_readfirstlane(tile_b);
    nsplit = __builtin_amdgcn_readfirstlane(nsplit);
    const int t = xcd_remap(tile_b, tiles_m * tiles_n);
    int tm, tn;
    if (p.col_strips) {
        // strips of 8 tile-columns, rows walked inside a strip: an XCD's contiguous chunk is then a range of
        // COLUMNS over all rows -> each B (weight) panel is pulled through the fabric by one XCD instead of eight
        constexpr int GN = 8;
        const int per_strip = GN * tiles_m;
        const int strip = t / per_strip, within = t % per_strip;
        const int cols_here = min(GN, tiles_n - strip * GN);
        tn = strip * GN + within % cols_here;
        tm = within / cols_here;
    } else {
        const int per_group = p.group_m * tiles_n;
        const int grp = t / per_group, within = t % per_group;
        const int rows_here = min(p.group_m, tiles_m - grp * p.group_m);
        tm = grp * p.group_m + within % rows_here;
        tn = within / rows_here;
    }
    const int m0 = tm * BM_EFF, n0 = tn * BN;

    const u32x4 ra = make_desc(p.A, p.a_bytes);
    const u32x4 rb = make_desc(p.B, p.b_bytes);
    const uint32_t smem_addr = lds_addr_of(smem);

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT_all = (p.K + BKT - 1) / BKT;
    const int kt0 = (int)(((long)KT_all * ks) / nsplit);
    const int KT = (p.debug & 2) ? 0 : (int)(((long)KT_all * (ks + 1)) / nsplit) - kt0;   // this block's K-tiles: kt0 .. kt0+KT-1
    auto stage = [&](int kt_local, int buf) {
        const int kt = kt0 + kt_local;
        // LDS layout: [A|B] per stage, except for the interleaved loop (PIPE 4): [A0][A1][B0][B1], so that the
        // stage index fits in the 16-bit immediate of its ds_reads
        const uint32_t sa = smem_addr + (PIPE >= 4 ? buf * A_BYTES : buf * (A_BYTES + B_BYTES));
        const uint32_t sb = smem_addr + (PIPE >= 4 ? NSTAGE * A_BYTES + buf * B_BYTES : buf * (A_BYTES + B_BYTES) + A_BYTES);
        stage_tile<BM, A_KMAJ, NT, BKT>(ra, sa, m0, kt * BKT, p.lda, tid);
        stage_tile<BN, B_KMAJ, NT, BKT>(rb, sb, n0, kt * BKT, p.ldb, tid);
    };
    auto compute = [&](int buf) {
        LDS_PTR(char) sa = smem + buf * (A_BYTES + B_BYTES);
        LDS_PTR(char) sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[j] = load_frag<BM, A_KMAJ, BKT>(sa, wm * WTM + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = load_frag<BN, B_KMAJ, BKT>(sb, wn * WTN + i * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (PIPE >= 7) {
        // ---- the three-stage loop below with a WEIGHT-ONLY FP8 B operand (round 4; SURVEY.md §8f item 4, BASELINE config 5) ----------
        // B = OCP e4m3fn codes [N][K] + one fp32 scale per output channel.  The codes are DMA'd as they are -- a K-tile of B is 16 KiB
        // instead of 32, two DMA pieces instead of four, 8-byte fragment reads instead of 16 -- and become bf16 MFMA operands on the
        // fragment path, in the shadow of the MFMAs of the preceding fragment:
        //   PIPE 7: v_cvt_pk_f32_fp8 -> v_pk_mul_f32 by the lane's row scale -> v_cvt_pk_bf16_f32: the operand is bf16(s * q), bit for
        //           bit what nv_fp8_dequant_rows writes (the semantics fixture G11 pins);
        //   PIPE 8: v_cvt_scalef32_pk_bf16_fp8 (one instruction per pair); p.fp8_epi = 0: the lane's scale is the instruction's scale
        //           operand, = 1: the codes are converted unscaled (exact: e4m3 fits bf16) and s[n] multiplies the fp32 accumulator in
        //           the epilogue (differs from bf16(s * q) by that one rounding per weight).
        // Why only here: the few-hundred-row GEMMs (K/V-reuse steps) are latency-bound -- MFMA pipe 37 % busy, the B-tile DMA and the
        // B fragment reads are the part of a K-step that does not shrink with the tile -- so halving B's bytes pays and
        // the conversions find idle issue slots; at prefill sizes the loop is MFMA-issue-bound and the pre-pass (3 B per weight) is
        // the cheaper form.
        // LDS image of a B stage: [256 rows][64 codes] = 64-B rows, four 16-B slots per row, phys slot = slot ^ ((row >> 2) & 3): the 16
        // rows x 2 halves a half-wave's ds_read_b64 touches cover the 64 banks once.
        static_assert(A_KMAJ && B_KMAJ && BKT == 64 && TN == 4 && NT == 512 && BN == 256 && TME >= 4 && TME <= 5, "fp8-B loop: cut-off tiles, forward layout");
        constexpr int A_STG = BM_EFF * BKT * 2;                 // 16 / 20 KiB
        constexpr int B_STG = BN * BKT;                         // 16 KiB of codes
        constexpr int B_REG = 3 * A_STG;
        constexpr int DUMMY = B_REG + 3 * B_STG;
        static_assert(DUMMY + 4096 <= 160 * 1024 && 2 * B_STG + 3 * 1024 + 64 < 65536, "LDS budget / ds_read immediates");
        constexpr int A_PCS = (BM_EFF + 63) / 64;
        constexpr bool HALF = (BM_EFF % 64) != 0;
        constexpr int B_PCS = B_STG / 8192;                     // 2
        constexpr int LT = A_PCS + B_PCS;
        FragAddr<BM, true, TM, WGM> fa_;
        fa_.init(wm * 16, lane);
        uint32_t offA[2], offB[2];
        const int idx_ = lane & 15, kg_ = lane >> 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            offA[q] = smem_addr + fa_.off[q];
            offB[q] = smem_addr + B_REG + (wn * WTN + idx_) * 64 + ((((q * 2 + (kg_ >> 1)) ^ ((idx_ >> 2) & 3))) << 4) + (kg_ & 1) * 8;
            asm volatile("" : "+v"(offA[q]));
            asm volatile("" : "+v"(offB[q]));
        }
        if (smem_addr != 0) __builtin_trap();
        auto ldA = [&](auto SC, auto JC, auto KC) -> bf16x8 {
            constexpr int st = decltype(SC)::value, j = decltype(JC)::value, kk = decltype(KC)::value;
            return *(LDS_PTR(bf16x8))(uintptr_t)(offA[kk] + (st * A_STG + j * 2048 * WGM));
        };
        auto ldB = [&](auto SC, auto IC, auto KC) -> u32x2 {       // the 8 codes k = kk*32 + kg*8 .. +7 of row wn*64 + i*16 + idx
            constexpr int st = decltype(SC)::value, i = decltype(IC)::value, kk = decltype(KC)::value;
            return *(LDS_PTR(u32x2))(uintptr_t)(offB[kk] + (st * B_STG + i * 1024));
        };
        // the lane's four row scales (row n0 + wn*64 + i*16 + idx): loaded by hand BEFORE the first DMA so that the prologue's counted
        // vmcnt wait covers them (loads retire in order) and no compiler-inserted vmcnt(0) drains the prefetch ring later
        float sc[TN];
        {
            const float* sp = p.b_scales + n0 + wn * WTN + idx_;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = n0 + wn * WTN + i * 16 + idx_;
                const float* q = sp + i * 16;
                if (n >= p.N) q = p.b_scales;                   // (rows past N read zero codes; any finite scale will do)
                asm volatile("global_load_dword %0, %1, off" : "=v"(sc[i]) : "v"(q) : "memory");
            }
        }
        auto cvt = [&](const u32x2& c, float s_) -> bf16x8 {
            if constexpr (PIPE == 7) {
                typedef __attribute__((ext_vector_type(2))) float f2;
                const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8(c[0], false) * s_, b = __builtin_amdgcn_cvt_pk_f32_fp8(c[0], true) * s_;
                const f2 d = __builtin_amdgcn_cvt_pk_f32_fp8(c[1], false) * s_, e = __builtin_amdgcn_cvt_pk_f32_fp8(c[1], true) * s_;
                const u32x4 v = {pack2bf(a[0], a[1]), pack2bf(b[0], b[1]), pack2bf(d[0], d[1]), pack2bf(e[0], e[1])};
                return __builtin_bit_cast(bf16x8, v);
            } else {
                const nv_bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(c[0], s_, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(c[0], s_, true);
                const nv_bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(c[1], s_, false), e = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(c[1], s_, true);
                const bf16x8 v = {a[0], a[1], b[0], b[1], d[0], d[1], e[0], e[1]};
                return v;
            }
        };
        // DMA addressing: A as in the bf16 loop; B: a piece = 128 rows x 64 B (8 KiB), lane -> row chunk*16 + lane/4, phys slot lane%4
        uint32_t pva = piece_voff<BM, true, NT, BKT>(m0, 0, p.lda, tid, 0);
        uint32_t pvb;
        {
            const int row = wave * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3);
            pvb = (uint32_t)((long)(n0 + row) * p.ldb + slot * 16);
        }
        asm volatile("" : "+v"(pva));
        asm volatile("" : "+v"(pvb));
        const uint32_t a_piece = (uint32_t)((NT / 64) * (1024 / (BKT * 2)) * 2) * (uint32_t)p.lda;
        const uint32_t b_piece = 128u * (uint32_t)p.ldb;
        const uint32_t wv = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t m0base = wv * 1024;
        uint32_t m0half[3];
#pragma unroll
        for (int st = 0; st < 3; ++st) m0half[st] = wv < 4 ? st * A_STG + 2 * 8192 + wv * 1024 : DUMMY + (wv - 4) * 1024;
        const uint32_t a_step = BKT * 2, b_step = BKT;
        uint64_t a_base = (uint64_t)p.A + (uint64_t)a_step * kt0, b_base = (uint64_t)p.B + (uint64_t)b_step * kt0;
        uint32_t a_left = p.a_bytes - a_step * (uint32_t)kt0, b_left = p.b_bytes - b_step * (uint32_t)kt0;
        auto piece = [&](auto SC, auto CC, const u32x4& da, const u32x4& db) {
            constexpr int st = decltype(SC)::value, c = decltype(CC)::value;
            if constexpr (c < A_PCS) {
                if constexpr (HALF && c == A_PCS - 1) set_m0_imm<0>(m0half[st]);
                else set_m0_imm<st * A_STG + c * 8192>(m0base);
                dma16_m0set(da, pva, a_piece * c);
            } else {
                set_m0_imm<B_REG + st * B_STG + (c - A_PCS) * 8192>(m0base);
                dma16_m0set(db, pvb, b_piece * (c - A_PCS));
            }
        };
        auto advance = [&]() { a_base += a_step; a_left -= a_step; b_base += b_step; b_left -= b_step; };
        static_for<3>([&](auto SC) {
            constexpr int st = decltype(SC)::value;
            if (st < KT) {
                const u32x4 da = make_desc((const void*)a_base, a_left), db = make_desc((const void*)b_base, b_left);
                static_for<LT>([&](auto CC) { piece(SC, CC, da, db); });
                advance();
            }
        });
        if (KT >= 3) wait_vmcnt<2 * LT>(); else if (KT == 2) wait_vmcnt<LT>(); else wait_vmcnt<0>();   // (covers the scale loads: issued first)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(sc[i]));     // the scales are valid from HERE on: no use may be scheduled above the wait
        const float cs[TN] = {(PIPE == 8 && p.fp8_epi) ? 1.f : sc[0], (PIPE == 8 && p.fp8_epi) ? 1.f : sc[1],
                              (PIPE == 8 && p.fp8_epi) ? 1.f : sc[2], (PIPE == 8 && p.fp8_epi) ? 1.f : sc[3]};
        bf16x8 fa0[TM], fa1[TM], fbc[TN];
        u32x2 cb0[TN], cb1[TN];
        using std::integral_constant;
        static_for<TME>([&](auto JC) { fa0[decltype(JC)::value] = ldA(integral_constant<int, 0>{}, JC, integral_constant<int, 0>{}); });
        static_for<TN>([&](auto IC) { cb0[decltype(IC)::value] = ldB(integral_constant<int, 0>{}, IC, integral_constant<int, 0>{}); });
        fbc[0] = cvt(cb0[0], cs[0]);
        constexpr int TOT = TN * TME;
        constexpr int NRD = TME + TN;
        // MFMA n of a phase (n = 0 .. TOT-1, in group n * 8 / TOT) multiplies B fragment i = n / TME, so fragment i is first needed in
        // group 2 i: it is converted in group 2 i - 1 of the SAME phase (i = 1 .. 3), and fragment 0 of the NEXT phase in group 7 of
        // this one (its codes were read in groups 3 - 5).
        auto body = [&](auto STC, bool in3, bool nx, bool w2) {
            constexpr int ST = decltype(STC)::value, SN = (ST + 1) % 3;
            using K0 = integral_constant<int, 0>;
            using K1 = integral_constant<int, 1>;
            static_for<8>([&](auto GC) {                        // phase 1: MFMAs of k-step 0, reads of k-step 1 (stage ST)
                constexpr int gq = decltype(GC)::value;
                static_for<NRD>([&](auto RC) {
                    constexpr int r = decltype(RC)::value;
                    if constexpr (r * 6 / NRD == gq) {
                        if constexpr (r < TME) fa1[r] = ldA(STC, RC, K1{});
                        else cb1[r - TME] = ldB(STC, integral_constant<int, r - TME>{}, K1{});
                    }
                });
                if constexpr (gq == 1) fbc[1] = cvt(cb0[1], cs[1]);
                if constexpr (gq == 3) fbc[2] = cvt(cb0[2], cs[2]);
                if constexpr (gq == 5) fbc[3] = cvt(cb0[3], cs[3]);
                static_for<TOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value;
                    if constexpr (n * 8 / TOT == gq) {
                        constexpr int i = n / TME, j = n % TME;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbc[i], fa0[j], acc[i][j], 0, 0, 0);
                    }
                });
                if constexpr (gq == 7) fbc[0] = cvt(cb1[0], cs[0]);
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (w2) wait_vmcnt<LT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 da = make_desc((const void*)a_base, a_left), db = make_desc((const void*)b_base, b_left);
            static_for<8>([&](auto GC) {                        // phase 2: MFMAs of k-step 1, tile kt+1's k-step 0 reads, DMA of tile kt+3
                constexpr int gq = decltype(GC)::value;
                if (nx) {
                    static_for<NRD>([&](auto RC) {
                        constexpr int r = decltype(RC)::value;
                        if constexpr (r * 6 / NRD == gq) {
                            if constexpr (r < TME) fa0[r] = ldA(integral_constant<int, SN>{}, RC, K0{});
                            else cb0[r - TME] = ldB(integral_constant<int, SN>{}, integral_constant<int, r - TME>{}, K0{});
                        }
                    });
                }
                if constexpr (gq == 1) fbc[1] = cvt(cb1[1], cs[1]);
                if constexpr (gq == 3) fbc[2] = cvt(cb1[2], cs[2]);
                if constexpr (gq == 5) fbc[3] = cvt(cb1[3], cs[3]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<TOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value;
                    if constexpr (n * 8 / TOT == gq) {
                        constexpr int i = n / TME, j = n % TME;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbc[i], fa1[j], acc[i][j], 0, 0, 0);
                    }
                });
                if constexpr (gq == 7) { if (nx) fbc[0] = cvt(cb0[0], cs[0]); }
                __builtin_amdgcn_sched_barrier(0);
                if (in3) {
                    if constexpr (gq < LT) piece(STC, GC, da, db);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (in3) advance();
        };
        int kt = 0;
        for (; kt + 5 < KT; kt += 3) {
            body(integral_constant<int, 0>{}, true, true, true);
            body(integral_constant<int, 1>{}, true, true, true);
            body(integral_constant<int, 2>{}, true, true, true);
        }
        for (; kt < KT; ++kt) {
            const bool in3 = kt + 3 < KT, nx = kt + 1 < KT, w2 = kt + 2 < KT;
            const int st = kt % 3;
            if (st == 0) body(integral_constant<int, 0>{}, in3, nx, w2);
            else if (st == 1) body(integral_constant<int, 1>{}, in3, nx, w2);
            else body(integral_constant<int, 2>{}, in3, nx, w2);
        }
        if (PIPE == 8 && p.fp8_epi) {
            // s[n] on the fp32 accumulators: the lane holds D[n = g*4 + r][m] per 16x16 tile, n = n0 + wn*64 + i*16 + g*4 + r
            const int g4 = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = n0 + wn * WTN + i * 16 + g4;
                f32x4 s4 = {1.f, 1.f, 1.f, 1.f};
                if (n + 3 < p.N) s4 = *(const f32x4*)(p.b_scales + n);
                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) s4[r] = p.b_scales[n + r]; }
#pragma unroll
                for (int j = 0; j < TMU; ++j) acc[i][j] *= s4;
            }
        }
    } else if constexpr (PIPE == 6) {
        // ---- THREE LDS stages for the cut-off tiles (TME = 4, 5: 128 / 160 x 256) ------------------------------------------------
        // Why: a K-tile of a cut-off tile takes ~0.6-0.7 us of MFMA time, and the two-stage loop below issues the DMA of tile kt+2
        // only one iteration before tile kt+1 is needed -- shorter than an L2-missing load takes on the few-hundred-row GEMMs these
        // tiles exist for, whose weight panels are shared by 4-5 tiles at most (rocprofv3 --pmc, M = 670, N = 12288, TME 5: waves
        // parked 52 % of their cycles, MFMA pipe 37 % busy, 2.7 TB/s from the fabric; profiles/r03_gemm_smallm_pmc.txt).  A 128 /
        // 160-row A image is 16 / 20 KiB, so three stages of (A + 32 KiB B) fit the 160 KiB LDS and the DMA of tile kt+3 goes out
        // TWO iterations ahead.  Same phase structure as the two-stage loop (one barrier per K-tile, fragments double-buffered,
        // every non-MFMA instruction slotted between MFMAs), but the loop is unrolled over the three stages so that the stage is a
        // compile-time immediate of every ds_read and of every M0 write (no address toggling), and the wait in front of the barrier
        // is a counted one: tile kt+2 may still be in flight.
        static_assert(A_KMAJ && BKT == 64 && TN == 4 && NT == 512 && BN == 256 && TME >= 4 && TME <= 5, "three-stage loop: cut-off tiles, K-major A");
        constexpr int A_STG = BM_EFF * BKT * 2;                 // 16 / 20 KiB
        constexpr int B_STG = B_BYTES;                          // 32 KiB
        constexpr int B_REG = 3 * A_STG;                        // the B stages follow the three A stages
        constexpr int DUMMY = B_REG + 3 * B_STG;                // 4 KiB: where the unused half of a half piece lands
        static_assert(DUMMY + 4096 <= 160 * 1024, "LDS budget");
        constexpr int A_PCS = (BM_EFF + 63) / 64;               // DMA pieces (64 rows each) of an A stage: 2, or 2 + a half
        constexpr bool HALF = (BM_EFF % 64) != 0;
        constexpr int LT = A_PCS + 4;                           // DMA instructions per wave and K-tile
        constexpr int NB = B_KMAJ ? 2 : TN;
        FragAddr<BM, true, TM, WGM> fa_;
        FragAddr<BN, B_KMAJ, TN> fb_;
        fa_.init(wm * 16, lane);
        fb_.init(wn * WTN, lane);
        uint32_t offA[2], offBlo[NB], offBhi[NB];               // B: stages 0/1 as immediates on `lo`, stage 2 on `hi` (16-bit ds offsets)
#pragma unroll
        for (int q = 0; q < 2; ++q) { offA[q] = smem_addr + fa_.off[q]; asm volatile("" : "+v"(offA[q])); }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            offBlo[q] = smem_addr + B_REG + fb_.off[q];
            offBhi[q] = offBlo[q] + 2 * B_STG;
            asm volatile("" : "+v"(offBlo[q]));
            asm volatile("" : "+v"(offBhi[q]));
        }
        if (smem_addr != 0) __builtin_trap();                   // the immediates below assume the dynamic LDS block starts at 0
        auto ldA = [&](auto SC, auto JC, auto KC) -> bf16x8 {
            constexpr int st = decltype(SC)::value, j = decltype(JC)::value, kk = decltype(KC)::value;
            return *(LDS_PTR(bf16x8))(uintptr_t)(offA[kk] + (st * A_STG + j * 2048 * WGM));
        };
        auto ldB = [&](auto SC, auto IC, auto KC) -> bf16x8 {
            constexpr int st = decltype(SC)::value, i = decltype(IC)::value, kk = decltype(KC)::value;
            constexpr int so = st < 2 ? st * B_STG : 0;
            if constexpr (B_KMAJ) {
                const uint32_t b = st < 2 ? offBlo[kk] : offBhi[kk];
                return *(LDS_PTR(bf16x8))(uintptr_t)(b + (so + i * 2048));
            } else {
                const uint32_t q = (st < 2 ? offBlo[i] : offBhi[i]) + (so + kk * (32 * BN * 2));
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)q);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(uintptr_t)(q + 4 * BN * 2));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                return __builtin_bit_cast(bf16x8, v);
            }
        };
        // DMA addressing (as in the two-stage loop): one per-lane source offset per operand, a scalar per piece, K advance in the
        // descriptors; destination M0 = this wave's 1 KiB slice + an immediate.  The half piece (rows 128..159 of a 160-row tile):
        // waves 0-3 carry it, waves 4-7 load the next 32 rows into the dummy area (keeps the per-wave load count uniform).
        uint32_t pva = piece_voff<BM, true, NT, BKT>(m0, 0, p.lda, tid, 0);
        uint32_t pvb = piece_voff<BN, B_KMAJ, NT, BKT>(n0, 0, p.ldb, tid, 0);
        asm volatile("" : "+v"(pva));
        asm volatile("" : "+v"(pvb));
        const uint32_t a_piece = (uint32_t)((NT / 64) * (1024 / (BKT * 2)) * 2) * (uint32_t)p.lda;
        const uint32_t b_piece = (uint32_t)((B_KMAJ ? (NT / 64) * (1024 / (BKT * 2)) : (NT / 64) * (1024 / (BN * 2))) * 2) * (uint32_t)p.ldb;
        const uint32_t wv = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t m0base = wv * 1024;
        uint32_t m0half[3];
#pragma unroll
        for (int st = 0; st < 3; ++st) m0half[st] = wv < 4 ? st * A_STG + 2 * 8192 + wv * 1024 : DUMMY + (wv - 4) * 1024;
        const uint32_t a_step = BKT * 2;
        const uint32_t b_step = B_KMAJ ? BKT * 2 : (uint32_t)(BKT * 2) * (uint32_t)p.ldb;
        uint64_t a_base = (uint64_t)p.A + (uint64_t)a_step * kt0, b_base = (uint64_t)p.B + (uint64_t)b_step * kt0;
        uint32_t a_left = p.a_bytes - a_step * (uint32_t)kt0, b_left = p.b_bytes - b_step * (uint32_t)kt0;
        auto piece = [&](auto SC, auto CC, const u32x4& da, const u32x4& db) {      // DMA instruction c (0..LT-1) of a tile into stage st
            constexpr int st = decltype(SC)::value, c = decltype(CC)::value;
            if constexpr (c < A_PCS) {
                if constexpr (HALF && c == A_PCS - 1) set_m0_imm<0>(m0half[st]);
                else set_m0_imm<st * A_STG + c * 8192>(m0base);
                dma16_m0set(da, pva, a_piece * c);
            } else {
                set_m0_imm<B_REG + st * B_STG + (c - A_PCS) * 8192>(m0base);
                dma16_m0set(db, pvb, b_piece * (c - A_PCS));
            }
        };
        auto advance = [&]() { a_base += a_step; a_left -= a_step; b_base += b_step; b_left -= b_step; };
        // prologue: tiles 0, 1, 2 into stages 0, 1, 2
        static_for<3>([&](auto SC) {
            constexpr int st = decltype(SC)::value;
            if (st < KT) {
                const u32x4 da = make_desc((const void*)a_base, a_left), db = make_desc((const void*)b_base, b_left);
                static_for<LT>([&](auto CC) { piece(SC, CC, da, db); });
                advance();
            }
        });
        if (KT >= 3) wait_vmcnt<2 * LT>(); else if (KT == 2) wait_vmcnt<LT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        using std::integral_constant;
        static_for<TME>([&](auto JC) { fa0[decltype(JC)::value] = ldA(integral_constant<int, 0>{}, JC, integral_constant<int, 0>{}); });
        static_for<TN>([&](auto IC) { fb0[decltype(IC)::value] = ldB(integral_constant<int, 0>{}, IC, integral_constant<int, 0>{}); });
        constexpr int TOT = TN * TME;                           // MFMAs per phase: 16 / 20, in 8 groups of 2-3
        constexpr int NRD = TME + TN;                           // fragment reads per phase: 8 / 9
        // one iteration = K-tile kt living in stage ST.  in3: tile kt+3 exists (DMA it into ST); nx: tile kt+1 exists (read its k-step
        // 0 fragments); w2: tile kt+2 exists, i.e. is still in flight at the barrier (counted wait)
        auto body = [&](auto STC, bool in3, bool nx, bool w2) {
            constexpr int ST = decltype(STC)::value, SN = (ST + 1) % 3;
            using K0 = integral_constant<int, 0>;
            using K1 = integral_constant<int, 1>;
            static_for<8>([&](auto GC) {                        // phase 1: MFMAs of k-step 0, reads of k-step 1 (stage ST)
                constexpr int gq = decltype(GC)::value;
                static_for<NRD>([&](auto RC) {                  // read r goes out in group r * 6 / NRD: all reads in the first 6 groups
                    constexpr int r = decltype(RC)::value;
                    if constexpr (r * 6 / NRD == gq) {
                        if constexpr (r < TME) fa1[r] = ldA(STC, RC, K1{});
                        else fb1[r - TME] = ldB(STC, integral_constant<int, r - TME>{}, K1{});
                    }
                });
                static_for<TOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value;
                    if constexpr (n * 8 / TOT == gq) {
                        constexpr int i = n / TME, j = n % TME;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[i], fa0[j], acc[i][j], 0, 0, 0);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (w2) wait_vmcnt<LT>(); else wait_vmcnt<0>();     // tile kt+1 has landed (this wave's share); kt+2 may be in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 da = make_desc((const void*)a_base, a_left), db = make_desc((const void*)b_base, b_left);
            static_for<8>([&](auto GC) {                        // phase 2: MFMAs of k-step 1, tile kt+1's k-step 0 reads, DMA of tile kt+3
                constexpr int gq = decltype(GC)::value;
                if (nx) {
                    static_for<NRD>([&](auto RC) {
                        constexpr int r = decltype(RC)::value;
                        if constexpr (r * 6 / NRD == gq) {
                            if constexpr (r < TME) fa0[r] = ldA(integral_constant<int, SN>{}, RC, K0{});
                            else fb0[r - TME] = ldB(integral_constant<int, SN>{}, integral_constant<int, r - TME>{}, K0{});
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<TOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value;
                    if constexpr (n * 8 / TOT == gq) {
                        constexpr int i = n / TME, j = n % TME;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[i], fa1[j], acc[i][j], 0, 0, 0);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                if (in3) {
                    if constexpr (gq < LT) piece(STC, GC, da, db);      // one DMA piece per group (LT <= 7 < 8 groups)
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (in3) advance();
        };
        int kt = 0;
        for (; kt + 5 < KT; kt += 3) {                          // all three tiles kt, kt+1, kt+2 have a tile three ahead
            body(integral_constant<int, 0>{}, true, true, true);
            body(integral_constant<int, 1>{}, true, true, true);
            body(integral_constant<int, 2>{}, true, true, true);
        }
        for (; kt < KT; ++kt) {                                 // the last (up to five) tiles; kt % 3 is still the stage
            const bool in3 = kt + 3 < KT, nx = kt + 1 < KT, w2 = kt + 2 < KT;
            const int st = kt % 3;
            if (st == 0) body(integral_constant<int, 0>{}, in3, nx, w2);
            else if (st == 1) body(integral_constant<int, 1>{}, in3, nx, w2);
            else body(integral_constant<int, 2>{}, in3, nx, w2);
        }
    } else if constexpr (PIPE >= 1) {
                // ---- software-pipelined main loop (2 LDS stages, BK=64 = 2 k-steps of 32) ----------------------
        // Fragment registers are double-buffered: the ds_reads of k-step 1 are in flight under the MFMAs of
        // k-step 0, and the ds_reads of the NEXT tile's k-step 0 under the MFMAs of k-step 1.  One barrier
        // per K-tile, in the middle: it publishes tile kt+1 (DMA'd a full iteration earlier) and retires every
        // wave's reads of tile kt's buffer, which the DMA of tile kt+2 then overwrites.
        static_assert(NSTAGE == 2 && BKT == 64, "pipelined loop is written for 2 stages of BK=64");
        constexpr int FSA = IL ? WGM : 1;                      // fragment-row stride of a wave along M
        FragAddr<BM, A_KMAJ, TM, FSA> fa_addr;
        FragAddr<BN, B_KMAJ, TN> fb_addr;
        fa_addr.init(IL ? wm * 16 : wm * WTM, lane);
        fb_addr.init(wn * WTN, lane);
        bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        auto ldfr = [&](int buf, int kk, bf16x8(&fa)[TM], bf16x8(&fb)[TN]) {
            LDS_PTR(char) sa = smem + (PIPE >= 4 ? buf * A_BYTES : buf * (A_BYTES + B_BYTES));
            LDS_PTR(char) sb = smem + (PIPE >= 4 ? NSTAGE * A_BYTES + buf * B_BYTES : buf * (A_BYTES + B_BYTES) + A_BYTES);
#pragma unroll
            for (int j = 0; j < TMU; ++j) fa[j] = fa_addr.load(sa, j, kk);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = fb_addr.load(sb, i, kk);
        };
        stage(0, 0);
        // counted wait only while nothing but DMA is in flight: a persistent block's later tiles still have the previous
        // tile's C stores outstanding, and loads/stores do not retire in order relative to each other
        if (KT > 1) { stage(1, 1); if (first) wait_vmcnt<LOADS>(); else wait_vmcnt<0>(); } else { wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
        ldfr(0, 0, fa0, fb0);
        static_assert(PIPE == 4, "the software-pipelined prologue above serves the hand-interleaved loop only");
        {
            // Hand-interleaved schedule: every non-MFMA instruction of a phase is slotted between groups of 4
            // MFMAs (an MFMA occupies the pipe for ~16 cycles but only one issue slot), so the matrix pipe never
            // waits for 12 ds_reads + 8 DMA issues to be pushed out first.  sched_barrier(0) after each group
            // keeps hipcc from regrouping them.
            // The steady-state loop is unrolled over the two stages and carries NO vector ALU work besides the MFMAs:
            // LDS addresses are per-stage VGPR constants + immediates (FragAddr2), the DMA source offsets are 2
            // loop-invariant VGPRs + scalar offsets, the K advance lives in the buffer descriptors (SALU: base += step, num_records -=
            // step, so the hardware bounds check still zero-fills ragged edges), the LDS destination of each DMA is
            // M0 = scalar base + immediate.  Ordinary VALU ops share the issue port with MFMA: the ~40 per K-step the
            // compiler-generated addressing cost were ~15% of the loop (tools/ubench/mix_rate.hip vs this kernel).
            static_assert(TM == 8 && TN == 4 && LOADS == 8, "interleave written for 128x64 wave tiles, 8 DMA/thread");
            constexpr int A_IT = A_BYTES / (NT * 16);
            // DMA pieces of the A image a cut-off tile needs: a K-major piece is 64 rows (all of them for an MN-major image, whose
            // pieces are k-rows)
            constexpr int A_IT_EFF = A_KMAJ ? (BM_EFF + 63) / 64 : A_IT;
            static_assert(NSTAGE == 2 && 2 * A_BYTES <= 65536 && 2 * B_BYTES <= 65536, "stage offset must fit the ds_read immediate");
            static_assert(TME >= 4, "the read/MFMA slots below assume at least the first four A fragments are live");
            FragAddr2<BM, A_KMAJ, TM, A_BYTES, WGM> fa2;
            FragAddr2<BN, B_KMAJ, TN, B_BYTES> fb2;
            fa2.init(wm * 16, lane, smem_addr);
            fb2.init(wn * WTN, lane, smem_addr + 2 * A_BYTES);
            // DMA source offsets: piece q of an operand = piece 0 + q * (a uniform number of bytes) -- the LDS swizzles
            // repeat every 8 KiB of image -- so ONE VGPR per operand plus a scalar offset per piece (the buffer bounds
            // check includes soffset on gfx950: tools/ubench/soffset_oob.hip).
            uint32_t pva = piece_voff<BM, A_KMAJ, NT, BKT>(m0, 0, p.lda, tid, 0);
            uint32_t pvb = piece_voff<BN, B_KMAJ, NT, BKT>(n0, 0, p.ldb, tid, 0);
            asm volatile("" : "+v"(pva));
            asm volatile("" : "+v"(pvb));
            // rows (K-major) or k-rows (MN-major) covered by one piece = 8 KiB of LDS image
            const uint32_t a_piece = (uint32_t)((A_KMAJ ? (NT / 64) * (1024 / (BKT * 2)) : (NT / 64) * (1024 / (BM * 2))) * 2) * (uint32_t)p.lda;
            const uint32_t b_piece = (uint32_t)((B_KMAJ ? (NT / 64) * (1024 / (BKT * 2)) : (NT / 64) * (1024 / (BN * 2))) * 2) * (uint32_t)p.ldb;
            const uint32_t m0base = __builtin_amdgcn_readfirstlane(smem_addr + wave * 1024);
            const uint32_t a_step = A_KMAJ ? BKT * 2 : (uint32_t)(BKT * 2) * (uint32_t)p.lda;   // bytes per K-tile
            const uint32_t b_step = B_KMAJ ? BKT * 2 : (uint32_t)(BKT * 2) * (uint32_t)p.ldb;
            uint64_t a_base = (uint64_t)p.A + (uint64_t)a_step * (kt0 + 2), b_base = (uint64_t)p.B + (uint64_t)b_step * (kt0 + 2);
            // bytes still addressable from the advanced base.  A DMA is only issued for K-tiles < KT_all, whose advance
            // is < the operand's span, so this never wraps while it is in use (plain SALU subtract, no clamp).
            uint32_t a_left = p.a_bytes - a_step * (uint32_t)(kt0 + 2), b_left = p.b_bytes - b_step * (uint32_t)(kt0 + 2);
            // K-major kk addressing: off[kk] of stage `cur`.  Phase 1 reads k-step 1 of the current stage, phase 2
            // reads k-step 0 of the other one: off[1] always points at the current stage, off[0] at the next.
            if (A_KMAJ) { fa2.off[0] ^= A_BYTES; } else { /* MN-major: one set of addresses, flipped twice per tile */ }
            if (B_KMAJ) { fb2.off[0] ^= B_BYTES; }
            uint32_t m0cur = m0base;                          // LDS address (wave's 1 KiB slice) of the current stage's A image
            const uint32_t m0sum = 2 * m0base + A_BYTES;
            if (smem_addr & 0xffffu) __builtin_trap();        // the xor stage toggles assume the dynamic LDS block starts 64 KiB-aligned (it starts at 0)
            auto body = [&](auto FULLC, bool more1, bool more2) {
                constexpr bool FULL = decltype(FULLC)::value;
                static_for<8>([&](auto CC) {                  // phase 1: MFMAs of k-step 0, loads of k-step 1
                    constexpr int c = decltype(CC)::value;
                    // the 12 fragment reads go out in the first 6 groups, so the last one has two groups of MFMAs
                    // to land before the lgkmcnt(0) + barrier below (and before the loop-top wait of the next tile)
                    if constexpr (c < 4) { fa1[c] = fa2.load(c, 1); fb1[c] = fb2.load(c, 1); }
                    else if constexpr (c < 6) {
                        if constexpr (2 * c - 4 < TME) fa1[2 * c - 4] = fa2.load(2 * c - 4, 1);
                        if constexpr (2 * c - 3 < TME) fa1[2 * c - 3] = fa2.load(2 * c - 3, 1);
                    }
                    static_for<4>([&](auto EC) {
                        constexpr int e = decltype(EC)::value;
                        constexpr int i = c >> 1, j = (c & 1) * 4 + e;
                        if constexpr (j < TME) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[i], fa0[j], acc[i][j], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (!A_KMAJ) fa2.flip();                      // MN-major: now address the next stage (k-step 0 reads)
                if (!B_KMAJ) fb2.flip();
                const u32x4 da = make_desc((const void*)a_base, a_left);
                const u32x4 db = make_desc((const void*)b_base, b_left);
                static_for<8>([&](auto CC) {                  // phase 2: MFMAs of k-step 1, next tile's k-step 0 + DMA
                    constexpr int c = decltype(CC)::value;
                    if (FULL || more1) {
                        if constexpr (c < 4) { fa0[c] = fa2.load(c, 0); fb0[c] = fb2.load(c, 0); }
                        else if constexpr (c < 6) {
                            if constexpr (2 * c - 4 < TME) fa0[2 * c - 4] = fa2.load(2 * c - 4, 0);
                            if constexpr (2 * c - 3 < TME) fa0[2 * c - 3] = fa2.load(2 * c - 3, 0);
                        }
                    }
                    // one DMA piece of tile kt+2 per group (bunching them earlier measured 3-8% slower); M0 is written
                    // before the group's MFMAs and consumed after them (no s_nop, the hazard distance is free)
                    constexpr bool has_dma = (c >= A_IT) || (c < A_IT_EFF);     // A pieces beyond a cut-off tile's rows are skipped
                    if constexpr (has_dma) {
                        if (FULL || more2) {
                            if constexpr (c < A_IT) set_m0_imm<c * (NT / 64) * 1024>(m0cur);
                            else set_m0_imm<2 * A_BYTES + (c - A_IT) * (NT / 64) * 1024>(m0cur);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<4>([&](auto EC) {
                        constexpr int e = decltype(EC)::value;
                        constexpr int i = c >> 1, j = (c & 1) * 4 + e;
                        if constexpr (j < TME) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[i], fa1[j], acc[i][j], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (has_dma) {
                        if (FULL || more2) {
                            if constexpr (c < A_IT) dma16_m0set(da, pva, a_piece * c);
                            else dma16_m0set(db, pvb, b_piece * (c - A_IT));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                a_base += a_step; a_left -= a_step;
                b_base += b_step; b_left -= b_step;
                m0cur = m0sum - m0cur;                        // other stage (A_BYTES == B_BYTES: one toggle serves both images)
                if (A_KMAJ) { fa2.flip(); }                   // K-major: both k-step addresses move to the other stage
                if (B_KMAJ) { fb2.flip(); }
            };
            static_assert(A_BYTES == B_BYTES, "one xor toggles both stage images");
            using std::integral_constant;
            int kt = 0;
            for (; kt + 2 < KT; ++kt) body(integral_constant<bool, true>{}, true, true);
            for (; kt < KT; ++kt) body(integral_constant<bool, false>{}, kt + 1 < KT, kt + 2 < KT);
        }
    } else {
    // ---- NSTAGE-deep DMA pipeline: stages kt+1 .. kt+NSTAGE-1 are in flight while kt is computed.
    // Each wave waits (counted vmcnt, never a drain in steady state) for ITS share of the next stage,
    // then the barrier at the top of the next iteration publishes every wave's share; the same barrier
    // retires the reads of the buffer that the new DMA overwrites.
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE - 1; ++s_)
        if (s_ < KT) stage(s_, s_);
    wait_tiles<LOADS, NSTAGE - 2>(min(NSTAGE - 1, KT) - 1);
    for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < KT) stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
        compute(kt % NSTAGE);
        const int inflight = min(NSTAGE - 1, KT - 1 - kt);     // stages issued and not yet needed... incl. kt+1
        if (inflight > 0) wait_tiles<LOADS, NSTAGE - 2>(inflight - 1);
    }
    }

    // ---- split-K tail: every slice publishes its fp32 partial and reduces ONE ROW BAND of the tile --------------------------
    // Placement-independent hand-off in the write-through form (cdna_hip_programming.md §6 G16 R1): sc1 slab stores go to memory
    // past the XCD's L2 (no release fence = no buffer_wbl2 of an L2 full of dirty C tiles, no acquire invalidate), every wave
    // drains its stores, one lane counts the slice in (relaxed agent-scope atomic) and reads are sc1 loads.  The slab image is the
    // accumulator register image (lane-linear 16-B accesses).
    // Round 3: the reduction is spread over the tile's slices.  Before, the LAST slice to arrive read all `nsplit` slabs (256 KiB
    // each at the ~65 GB/s one CU gets from remote memory: 12-20 us with 3-5 slices) while the other slices' CUs had left.  Now
    // slice s owns the fragment rows j in [TMU*s/n, TMU*(s+1)/n) of every wave -- a contiguous band of tile rows, thanks to the
    // interleaved wave rows -- stores its whole partial, waits until all n
    // slices are in (they sit within 8*n consecutive blocks of the dispatch order, see above; the spin is bounded and traps rather
    // than hangs) and sums its band over the slices IN SLICE ORDER 0..n-1 (fp32 addition is not associative: an arrival-order sum
    // made results depend on block timing, tests/test_round2_gpu.py::test_full_vicuna_7b_training_step_invariants), then runs the
    // epilogue for its band only.  Per slice: one slab written, one slab's worth read (n bands of 1/n), all CUs of the round busy.
    int j_lo = 0, j_hi = TMU;                              // fragment rows (per wave) this block finishes and stores
    if (nsplit > 1) {
        constexpr int SLAB = BM * BN;
        LDS_PTR(unsigned) flag = (LDS_PTR(unsigned))smem;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.slabs, 0, 0x7fffffff, 0x00020000);
        const bool band = IL && p.band_reduce;
        if (band) { j_lo = (TMU * ks) / nsplit; j_hi = (TMU * (ks + 1)) / nsplit; }
        const int mine = (tail_u * p.split + ks) * SLAB * 4;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TMU; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                           mine + ((((wave * TN + i) * TM + j) * 64 + lane) * 16), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (band) {
            if (tid == 0) {
                __hip_atomic_fetch_add(p.counters + tail_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (j_hi > j_lo) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.counters + tail_u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nsplit) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 22)) __builtin_trap();      // ~seconds: a partner slice never ran -- fail loudly, never hang the GPU
                    }
                }
            }
            __syncthreads();
            if (j_hi > j_lo) {
                for (int o = 0; o < nsplit; ++o) {             // slice order (deterministic sum), own slab included: same lane, same address
                    const int part = (tail_u * p.split + o) * SLAB * 4;
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TMU; ++j) {
                            if (j < j_lo || j >= j_hi) continue;
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, part + ((((wave * TN + i) * TM + j) * 64 + lane) * 16), 0, 16));
                            acc[i][j] = (o == 0) ? v : acc[i][j] + v;
                        }
                }
            }
            // count out; the last slice to leave zeroes both words for the next launch (everyone has seen `nsplit` by then)
            if (tid == 0) {
                const unsigned gone = __hip_atomic_fetch_add(p.counters + 512 + tail_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (gone == (unsigned)(nsplit - 1)) {
                    __hip_atomic_store(p.counters + tail_u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.counters + 512 + tail_u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (j_hi <= j_lo) return;                          // more slices than fragment rows: this one owns no band
        } else {
            // the round-2 form (NV_GEMM_BAND_REDUCE=0, and the 128x128 / non-interleaved tiles): the last arriver reduces every slab
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.counters + tail_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned ticket = *flag;
            if (ticket != (unsigned)(nsplit - 1)) return;
            if (tid == 0) __hip_atomic_store(p.counters + tail_u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            for (int o = 0; o < nsplit; ++o) {                 // slice order, whichever block reduces (deterministic sum)
                const int part = (tail_u * p.split + o) * SLAB * 4;
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TMU; ++j) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, part + ((((wave * TN + i) * TM + j) * 64 + lane) * 16), 0, 16));
                        acc[i][j] = (o == 0) ? v : acc[i][j] + v;
                    }
            }
        }
    }

    // ---- epilogue: lane holds D[n = g*4+r][m = lane&15] per 16x16 tile ----
    if (p.debug & 1) return;
    const int g = lane >> 4, mi = lane & 15;
    // Staged path: the C tile goes through LDS so that the global side is row-contiguous -- one store instruction
    // covers 1 KiB of whole 128-B lines (2 rows x 512 B for BN=256) instead of 16 rows x 32 B straight from the MFMA
    // register image.  Measured on MI355X (tools/ubench/store_rate.hip): 2.4 us vs 6.9 us per round of 256 tiles.
    // LDS image: [BM][BN] bf16, 16-B slot s of row m at s ^ (m & 15)  (conflict-free for the 8-B transposed writes
    // and the 16-B row reads).  Needs 16-B aligned rows; anything else takes the element-wise path below.
    static_assert(BM * BN * 2 <= NSTAGE * (BM + BN) * BKT * 2, "C tile image must fit in the stage buffers");
    const bool staged = ((p.ldc & 7) == 0) && ((p.N & 7) == 0) && ((((uintptr_t)p.C) & 15) == 0) &&
                        ((EPI != EPI_RESID && EPI != EPI_SWIGLU_BWD) || (((p.ldr & 7) == 0) && ((((uintptr_t)p.R) & 15) == 0))) &&
                        (EPI != EPI_ROPE || (BN % 128 == 0)) &&
                        (EPI != EPI_BIAS || ((((uintptr_t)p.R) & 7) == 0));
    if (staged) {
        __syncthreads();                                  // every wave is done with the operand stages
#pragma unroll
        for (int j = 0; j < TMU; ++j) {
            if (j < j_lo || j >= j_hi) continue;              // (split-K band reduction: only this block's band is final)
            const int ml = IL ? (j * WGM + wm) * 16 + mi : wm * WTM + j * 16 + mi;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int nl = wn * WTN + i * 16 + g * 4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (EPI == EPI_BIAS) {
                    const int n = n0 + nl;
                    if (n < p.N) {                        // N % 8 == 0 and n % 4 == 0: the 4 columns are in or out together
                        const u32x2 bb = *(const u32x2*)(p.R + n);
                        v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
                        v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
                    }
                }
                const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                *(LDS_PTR(u32x2))(smem + ml * (BN * 2) + ((((nl >> 3) ^ (ml & 15))) << 4) + ((nl >> 2) & 1) * 8) = o;
            }
        }
        __syncthreads();
        constexpr int SLOTS = BN / 8;                     // 16-B slots per C row
        constexpr int ROWS_PER_PASS = NT / SLOTS;
        static_assert(!IL || (WGM * 16) % ROWS_PER_PASS == 0, "a band of fragment rows must be whole store passes");
        const int c16 = tid % SLOTS, r_in = tid / SLOTS;
        const int n = n0 + c16 * 8;
        if (n < p.N) {
            // rows of the band: fragment rows [j_lo*WGM, j_hi*WGM) of the tile (interleaved wave rows) = a contiguous row range
            const int pass_lo = IL ? (j_lo * WGM * 16) / ROWS_PER_PASS : 0, pass_hi = IL ? (j_hi * WGM * 16) / ROWS_PER_PASS : BM_EFF / ROWS_PER_PASS;
            // Round 5: the epilogues that READ global memory per C row -- the residual / the gradient being accumulated into, the RoPE
            // position and its cos / sin rows -- issue those loads for EIGHT passes up front and only then walk the passes.  C, R and
            // the tables are plain pointers of one struct: hipcc has to assume that a store to C may alias the next pass's load, so the
            // one-pass-at-a-time loop below paid a full memory latency per pass (measured through the accumulate epilogue: the same
            // weight-gradient GEMMs ran 1 249 TF accumulating into zeros and 1 320 TF storing; in the forward, the plain-store gate|up
            // GEMM 1 403 TF against 1 240 / 1 267 for the residual / RoPE ones).  Same arithmetic, same order per element.
            if constexpr (EPI == EPI_ACCUM || EPI == EPI_RESID || EPI == EPI_ROPE) if (p.epi_preload) {
                constexpr int CH = 8;
                const bool roped = (EPI == EPI_ROPE) && (n < p.rope_cols);
                const int c = n & 127;                                 // (RoPE) 8 columns c .. c+7 inside one half of a head
                const bool lo = c < 64;
                const int pc16 = (c16 + (lo ? 8 : -8));                 // partner 16-B slot (64 columns away)
                const float sgn = lo ? -1.f : 1.f;
                for (int p0 = pass_lo; p0 < pass_hi; p0 += CH) {
                    u32x4 ra[CH], rb[CH];
                    if (EPI == EPI_ROPE) {
                        if (roped) {
                            int posq[CH];
#pragma unroll
                            for (int q = 0; q < CH; ++q) {
                                const int m = m0 + (p0 + q) * ROWS_PER_PASS + r_in;
                                const bool ok = (p0 + q < pass_hi) && (m < p.M);
                                posq[q] = ok ? (p.rope_pos ? p.rope_pos[m] : m % p.rope_S) : 0;
                            }
#pragma unroll
                            for (int q = 0; q < CH; ++q) {
                                ra[q] = *(const u32x4*)(p.R + (long)posq[q] * 128 + (c & 63));
                                rb[q] = *(const u32x4*)(p.rope_sin + (long)posq[q] * 128 + (c & 63));
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < CH; ++q) {
                            const int m = m0 + (p0 + q) * ROWS_PER_PASS + r_in;
                            if ((p0 + q < pass_hi) && (m < p.M))
                                ra[q] = *(const u32x4*)((EPI == EPI_ACCUM) ? p.C + (long)m * p.ldc + n : p.R + (long)m * p.ldr + n);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int pass = p0 + q;
                        const int ml = pass * ROWS_PER_PASS + r_in, m = m0 + ml;
                        if (pass >= pass_hi || m >= p.M) break;
                        u32x4 t = *(LDS_PTR(u32x4))(smem + ml * (BN * 2) + ((c16 ^ (ml & 15)) << 4));
                        bf16_t* cp = p.C + (long)m * p.ldc + n;
                        if (EPI == EPI_ROPE) {
                            // head_dim 128, rotate-half: column c of a head pairs with c +- 64 -- the same row of the LDS image
                            // (a 256-wide tile holds two whole heads).  Arithmetic = rope_kernel (lm_rowops.hip), bit for bit.
                            if (roped) {
                                const u32x4 pr = *(LDS_PTR(u32x4))(smem + ml * (BN * 2) + ((pc16 ^ (ml & 15)) << 4));
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float x0 = __uint_as_float(t[e] << 16), x1 = __uint_as_float(t[e] & 0xffff0000u);
                                    const float y0 = __uint_as_float(pr[e] << 16), y1 = __uint_as_float(pr[e] & 0xffff0000u);
                                    const float c0 = __uint_as_float(ra[q][e] << 16), c1 = __uint_as_float(ra[q][e] & 0xffff0000u);
                                    const float s0 = __uint_as_float(rb[q][e] << 16), s1 = __uint_as_float(rb[q][e] & 0xffff0000u);
                                    t[e] = pack2bf(rbf(x0 * c0) + rbf(sgn * y0 * s0), rbf(x1 * c1) + rbf(sgn * y1 * s1));
                                }
                            }
                        } else {                                       // torch: out = resid + bf16(acc)  /  grad += bf16(dW)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                t[e] = pack2bf(__uint_as_float(ra[q][e] << 16) + __uint_as_float(t[e] << 16),
                                               __uint_as_float(ra[q][e] & 0xffff0000u) + __uint_as_float(t[e] & 0xffff0000u));
                        }
                        store_c(cp, t, p.c_nt);
                    }
                }
                return;
            }
#pragma unroll 4
            for (int pass = pass_lo; pass < pass_hi; ++pass) {
                const int ml = pass * ROWS_PER_PASS + r_in, m = m0 + ml;
                if (m >= p.M) break;
                u32x4 t = *(LDS_PTR(u32x4))(smem + ml * (BN * 2) + ((c16 ^ (ml & 15)) << 4));
                bf16_t* cp = p.C + (long)m * p.ldc + n;
                if (EPI == EPI_ACCUM || EPI == EPI_RESID) {    // torch: out = resid + bf16(acc)  /  grad += bf16(dW)
                    const u32x4 rr = *(const u32x4*)((EPI == EPI_ACCUM) ? cp : p.R + (long)m * p.ldr + n);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        t[q] = pack2bf(__uint_as_float(rr[q] << 16) + __uint_as_float(t[q] << 16),
                                       __uint_as_float(rr[q] & 0xffff0000u) + __uint_as_float(t[q] & 0xffff0000u));
                }
                if (EPI == EPI_ROPE) {
                    // head_dim 128, rotate-half: column c of a head pairs with c +- 64 -- the same row of the LDS image
                    // (a 256-wide tile holds two whole heads).  Arithmetic = rope_kernel (lm_rowops.hip), bit for bit.
                    if (n < p.rope_cols) {
                        const int c = n & 127;                         // 8 columns c .. c+7 inside one half
                        const bool lo = c < 64;
                        const int pc16 = (c16 + (lo ? 8 : -8));         // partner 16-B slot (64 columns away)
                        const u32x4 pr = *(LDS_PTR(u32x4))(smem + ml * (BN * 2) + ((pc16 ^ (ml & 15)) << 4));
                        const int pos = p.rope_pos ? p.rope_pos[m] : m % p.rope_S;
                        const u32x4 cw = *(const u32x4*)(p.R + (long)pos * 128 + (c & 63));
                        const u32x4 sw = *(const u32x4*)(p.rope_sin + (long)pos * 128 + (c & 63));
                        const float sgn = lo ? -1.f : 1.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x0 = __uint_as_float(t[q] << 16), x1 = __uint_as_float(t[q] & 0xffff0000u);
                            const float y0 = __uint_as_float(pr[q] << 16), y1 = __uint_as_float(pr[q] & 0xffff0000u);
                            const float c0 = __uint_as_float(cw[q] << 16), c1 = __uint_as_float(cw[q] & 0xffff0000u);
                            const float s0 = __uint_as_float(sw[q] << 16), s1 = __uint_as_float(sw[q] & 0xffff0000u);
                            t[q] = pack2bf(rbf(x0 * c0) + rbf(sgn * y0 * s0), rbf(x1 * c1) + rbf(sgn * y1 * s1));
                        }
                    }
                    store_c(cp, t, p.c_nt);
                    continue;
                }
                if (EPI == EPI_SWIGLU_BWD) {
                    // C = d(gate|up) [M, 2N]: the tile's dh never goes to HBM; gate/up come in 16 B per lane like a residual
                    const bf16_t* gp = p.R + (long)m * p.ldr + n;
                    const u32x4 gg = *(const u32x4*)gp, uu = *(const u32x4*)(gp + p.N);
                    u32x4 og, ou;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float dg0, du0, dg1, du1;
                        swiglu_bwd_elem(__uint_as_float(t[q] << 16), __uint_as_float(gg[q] << 16), __uint_as_float(uu[q] << 16), dg0, du0);
                        swiglu_bwd_elem(__uint_as_float(t[q] & 0xffff0000u), __uint_as_float(gg[q] & 0xffff0000u),
                                        __uint_as_float(uu[q] & 0xffff0000u), dg1, du1);
                        og[q] = pack2bf(dg0, dg1);
                        ou[q] = pack2bf(du0, du1);
                    }
                    *(u32x4*)cp = og;
                    *(u32x4*)(cp + p.N) = ou;
                    continue;
                }
                store_c(cp, t, p.c_nt);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TMU; ++j) {
        if (j < j_lo || j >= j_hi) continue;
        const int m = m0 + (IL ? (j * WGM + wm) * 16 + mi : wm * WTM + j * 16 + mi);
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WTN + i * 16 + g * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            bf16_t* cp = p.C + (long)m * p.ldc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= p.N) continue;
                if (EPI == EPI_BIAS) v[r] += bf2f(p.R[n + r]);
                else if (EPI == EPI_ACCUM) v[r] = bf2f(cp[r]) + rbf(v[r]);   // torch: grad += bf16(dW)
                else if (EPI == EPI_RESID) v[r] = bf2f(p.R[(long)m * p.ldr + n + r]) + rbf(v[r]);
                else if (EPI == EPI_SWIGLU_BWD) {
                    float dg, du;
                    swiglu_bwd_elem(rbf(v[r]), bf2f(p.R[(long)m * p.ldr + n + r]), bf2f(p.R[(long)m * p.ldr + p.N + n + r]), dg, du);
                    cp[r] = f2bf(dg);
                    cp[p.N + r] = f2bf(du);
                    continue;
                }
                cp[r] = f2bf(v[r]);
            }
        }
    }
}

// ---- host-side launch planning ------------------------------------------------------------------------------------------
// Split-K tail: with one 256x256 block per CU, T tiles run in ceil(T/256) rounds and the last round is often nearly empty
// (M=5152, N=4096: 336 tiles = 2 rounds for 1.31 rounds of work).  The tiles of a partial round that is at most half full are
// cut into `split` K-slices so the round is ~full and 1/split as long.  Slices of >= ~20 K-tiles: shorter ones are dominated by
// their prologue + slab hand-off (M=4744, tools/gemm_probe.py --cold: K=4096 o_proj forward / dgrad 893 / 852 TF with 4 slices
// of 16, 957 / 894 with 3; the 344-K-tile gate|up dgrad 1198 with 4 slices, 1229 with 5).  NV_GEMM_MAXSPLIT / NV_GEMM_MINSLICE
// are measurement knobs.  (Measured and removed in round 2/3: two uneven slices for a 50-80 % full round -- 3.5-8 % slower -- and
// dispatching the tail slices first.)
inline int tail_split(int rem, int KT) {
    constexpr int CUS = 256;
    if (rem <= 0 || rem > CUS / 2) return 1;
    static const int max_split_env = [] { const char* e = getenv("NV_GEMM_MAXSPLIT"); return e ? atoi(e) : 0; }();
    static const int min_slice_env = [] { const char* e = getenv("NV_GEMM_MINSLICE"); return e ? atoi(e) : 0; }();
    // a handful of tail tiles (M = 670, N = 22016: 258 tiles -> 2) may be cut finer: few slabs in flight, and the round they
    // occupy is otherwise empty (round 3, same probe: 136 -> 126 us); a 48-tile tail is not (o_proj at M = 4760: 137 -> 144 us)
    const int max_split = max_split_env ? max_split_env : (rem <= 16 ? 8 : 6);
    const int min_slice = min_slice_env ? min_slice_env : (rem <= 16 ? 8 : 20);
    int split = CUS / rem;
    if (split > max_split) split = max_split;
    if (split > KT / min_slice) split = KT / min_slice;
    if (split * rem > MAX_SLABS) split = MAX_SLABS / rem;
    return split >= 2 ? split : 1;
}

// Estimated duration (us) of the 256-wide tile kernel with TME fragment rows per wave (tile = 32*TME x 256) on an M x N x K
// problem: full rounds of 256 tiles at t_k(TME) per 64-deep K-step, then the last partial round -- whole, at a K-step time that
// shrinks a little with the fraction of CUs it occupies, or as split-K slices (slower K-steps: the slices of a tile share no
// operand panels) plus the slab hand-off.  The constants are a least-squares fit (tools/fit_tme_model.py) to the durations
// tools/gemm_tme_probe.py measured on MI355X for the four Linear shapes of Vicuna-7B at M = 500 .. 5134, forward (NT) and dgrad
// (NN) layouts, TME = 4 .. 8 (profiles/r03_gemm_tme_probe_v3_three_stage.txt): rms error 7-8 %.  What the fit says about the
// kernel: a K-step of the two-stage loop costs ~0.6 us + 0.11 us per fragment row (1.38 / 1.48 us at TME = 7 / 8: the barrier, the
// B-tile DMA and the B fragment reads do not shrink with the tile), so its cut-off tiles only pay where they remove a mostly-empty
// round or a mostly-padding tile row; the three-stage loop of TME = 4 / 5 runs a K-step in 0.83 / 1.08 us (two-stage: 1.03 / 1.15),
// i.e. the 128-row tile is within 13 % of the full tile's time per flop.  With the band-distributed split-K reduction the
// hand-off term fell from ~15 us per slice to ~4.
struct TmeModel { double tk[5], oh0, c0, fix0, fix1, kfrac; };
inline const TmeModel& tme_model(bool b_kmaj) {
    static const TmeModel nt{{0.835, 1.080, 1.229, 1.380, 1.480}, 1.74, 0.799, -2.4, 3.8, 1.393};
    static const TmeModel nn{{0.834, 1.081, 1.247, 1.409, 1.523}, 0.0, 0.734, 5.1, 3.5, 1.219};
    return b_kmaj ? nt : nn;
}
inline double est_us_256(int M, int N, int K, int tme, bool can_split, bool b_kmaj) {
    const TmeModel& c = tme_model(b_kmaj);
    const int bme = 32 * tme, KT = (K + 63) / 64;
    const long T = (long)((M + bme - 1) / bme) * ((N + 255) / 256);
    const double tk = c.tk[tme - 4], oh = c.oh0 * (0.5 + 0.5 * tme / 8.0);
    const long full = T / 256;
    const int rem = (int)(T % 256);
    double t = full * (KT * tk + oh);
    if (rem) {
        const int split = can_split ? tail_split(rem, KT) : 1;
        double f = rem * split / 256.0;
        if (f > 1.0) f = 1.0;
        const double tke = tk * (c.c0 + (1.0 - c.c0) * f);
        if (split >= 2) {
            double fix = c.fix0 + c.fix1 * split;
            if (fix < 4.0) fix = 4.0;
            t += (KT / (double)split) * tke * c.kfrac + oh + fix;
        } else {
            t += KT * tke + oh;
        }
    }
    return t;
}
// 128x128 tile, two blocks per CU (tile_cfg 1), same probe: ~0.63 us per K-step while at most one block sits on a CU, 0.95 (NT) /
// 0.80 (NN) with two; never with a long contraction (K >= 8192: 0.87-0.89 us per K-step even at 128 tiles)
inline double est_us_128(int M, int N, int K, bool b_kmaj) {
    if (K >= 8192) return 1e30;
    const long T = (long)((M + 127) / 128) * ((N + 127) / 128);
    const int KT = (K + 63) / 64;
    const long rounds = (T + 511) / 512;
    return rounds * (KT * (T > 256 ? (b_kmaj ? 0.95 : 0.80) : 0.63) + 6.0);
}
// fragment rows per wave for this problem: the candidate with the smallest estimate; a cut-off tile must win by 3 %.
// NV_GEMM_TME = 4..8 forces one (measurement), NV_GEMM_TME = 8 therefore restores the round-2 behaviour.
inline int plan_tme(int M, int N, int K, bool can_split, bool b_kmaj, double* best_us) {
    static const int forced = [] { const char* e = getenv("NV_GEMM_TME"); return e ? atoi(e) : 0; }();
    if (forced >= 4 && forced <= 8) {
        if (best_us) *best_us = est_us_256(M, N, K, forced, can_split, b_kmaj);
        return forced;
    }
    int best = 8;
    double tb = est_us_256(M, N, K, 8, can_split, b_kmaj);
    const double t8 = tb;
    for (int t = 7; t >= 4; --t) {
        const double e = est_us_256(M, N, K, t, can_split, b_kmaj);
        if (e < 0.97 * t8 && e < tb) { best = t; tb = e; }
    }
    if (best_us) *best_us = tb;
    return best;
}

// Optional persistent form (NV_GEMM_PERSIST=1; off by default): the grid is at most one block per CU and each block walks
// its work items (item, item + grid, ...; 256 % 8 == 0, so an item keeps the XCD its id implies); the C stores of tile i
// drain while the first K-tiles of tile i+1 are already on their way into LDS.  Measured on MI355X: identical to the
// hardware dispatcher placing one block per tile (1306 vs 1306 TFLOP/s at 5152x12288x4096) -- the ~10 us per round
// of tiles that is not K-loop is prologue/epilogue latency inside the tile, not dispatch or store drain.
template <int BM, int BN, int WGM, int WGN, int BKT, int NSTAGE, bool A_KMAJ, bool B_KMAJ, int EPI, int PIPE, int TME>
__global__ __launch_bounds__(WGM* WGN * 64) __attribute__((amdgpu_waves_per_eu(1, ((NSTAGE * (BM + BN) * BKT * 2 > 80 * 1024) ? 1 : 2) * (WGM * WGN) / 4)))
void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    unsigned long long c0 = 0, w0 = 0;
    if (p.debug & 4) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
        const bool first = item == (int)blockIdx.x;
        if (!first) __syncthreads();          // every wave is done with the LDS image of the previous tile's C
        gemm_tile<BM, BN, WGM, WGN, BKT, NSTAGE, A_KMAJ, B_KMAJ, EPI, PIPE, TME>(p, item, first, smem);
    }
    // NV_GEMM_DEBUG bit 2 (measurement): block 0 leaves its core-clock cycles and 100 MHz wall ticks in the workspace
    // (bytes 4080..4095) -> average shader clock of the launch = 0.1 GHz * cycles / ticks
    if ((p.debug & 4) && blockIdx.x == 0 && threadIdx.x == 0 && p.counters) {
        unsigned long long* o = (unsigned long long*)((char*)p.counters + 4080);
        o[0] = __builtin_readcyclecounter() - c0;
        o[1] = wall_clock64() - w0;
    }
}

template <int BM, int BN, int WGM, int WGN, int BKT, int NSTAGE, bool A_KMAJ, bool B_KMAJ, int EPI, int PIPE = 0, int TME = BM / WGM / 16>
int launch(const GemmArgs& p, hipStream_t st) {
    constexpr int BM_EFF = (PIPE == 4 || PIPE >= 6) ? WGM * TME * 16 : BM;
    // the three-stage loop of the cut-off tiles: 3 x (A image of BM_EFF rows + B image) + a 4 KiB dummy landing area (fp8 B: 1 byte per
    // weight; at least the C tile's staging image, BM_EFF rows of 512 B)
    constexpr int LDS3 = (PIPE >= 7) ? 3 * (BM_EFF * BKT * 2 + BN * BKT) + 4096 : 3 * (BM_EFF + BN) * BKT * 2 + 4096;
    constexpr int LDS = (PIPE >= 6) ? (LDS3 > BM_EFF * BN * 2 ? LDS3 : BM_EFF * BN * 2) : NSTAGE * (BM + BN) * BKT * 2;
    auto kern = gemm_bf16_kernel<BM, BN, WGM, WGN, BKT, NSTAGE, A_KMAJ, B_KMAJ, EPI, PIPE, TME>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return NV_ERR_LAUNCH;
        attr_done = true;
    }
    const int tiles = ((p.M + BM_EFF - 1) / BM_EFF) * ((p.N + BN - 1) / BN);
    GemmArgs q = p;
    q.full_blocks = tiles; q.rem = 1; q.split = 1;
    // Split-K tail: with one 256x256 block per CU, T tiles run in ceil(T/256) rounds and the last round is
    // often nearly empty (M=5152,N=4096: 336 tiles = 2 rounds for 1.31 rounds of work).  The tiles of that
    // partial round are cut into `split` K-slices so the round is ~full and 1/split as long.
    constexpr int CUS = 256;
    if (BM == 256 && BN == 256 && p.slabs && p.counters) {
        const int rem = tiles % CUS, KT = (p.K + BKT - 1) / BKT;
        const int split = tail_split(rem, KT);
        if (split >= 2) { q.full_blocks = tiles - rem; q.rem = rem; q.split = split; }
    }
    // tail items: per XCD (item % 8) the slices of its chunk of tail tiles, padded to whole groups of 8 (see gemm_tile)
    const int ntail = q.split > 1 ? 8 * ((q.rem + 7) / 8) * q.split : 0;
    q.items = q.full_blocks + ntail;
    // persistent blocks for the interleaved kernel (measurement knob NV_GEMM_PERSIST; never with a split tail: its slices wait for
    // one another and must all be resident)
    const int grid = (PIPE == 4 && p.persist && ntail == 0 && q.items > CUS) ? CUS : q.items;
    NV_LAUNCH(kern, dim3(grid), dim3(WGM * WGN * 64), LDS, st, q);
    return nv_check_launch();
}

// cut-off tiles (TME < 8) are instantiated for the GEMMs of few-hundred-row steps (K/V-reuse inference, the suffix steps of
// navillm_amd/episode.py) and of the forward pass: y = x W^T with the store / residual / RoPE epilogues, and the dgrad dx = dy W
template <bool A_KMAJ, bool B_KMAJ, int EPI>
constexpr bool tme_instances = (A_KMAJ && B_KMAJ && (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_ROPE)) || (A_KMAJ && !B_KMAJ && EPI == EPI_STORE);

template <bool A_KMAJ, bool B_KMAJ, int EPI>
int launch_tme(const GemmArgs& p, int tme, hipStream_t st, bool two_stage = false) {
    if constexpr (tme_instances<A_KMAJ, B_KMAJ, EPI>) {
        // TME 4 and 5 run the three-stage loop (NV_GEMM_3STAGE=0 or tile_cfg 94 / 95: the two-stage one, for A/B measurements)
        static const int three = [] { const char* e = getenv("NV_GEMM_3STAGE"); return e ? atoi(e) : 1; }();
        if (three && !two_stage) {
            if (tme == 4) return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 6, 4>(p, st);
            if (tme == 5) return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 6, 5>(p, st);
        }
        switch (tme) {
            case 4: return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 4, 4>(p, st);
            case 5: return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 4, 5>(p, st);
            case 6: return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 4, 6>(p, st);
            case 7: return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 4, 7>(p, st);
        }
    }
    return launch<256, 256, 2, 4, 64, 2, A_KMAJ, B_KMAJ, EPI, 4, 8>(p, st);
}

// weight-only fp8 B operand: the three-stage cut-off tiles only (PIPE 7: exact bf16(s*q) operands, PIPE 8: v_cvt_scalef32)
template <int EPI>
int launch_fp8(const GemmArgs& p, int tme, int pipe, hipStream_t st) {
    if (pipe == 7) {
        if (tme == 4) return launch<256, 256, 2, 4, 64, 2, true, true, EPI, 7, 4>(p, st);
        if (tme == 5) return launch<256, 256, 2, 4, 64, 2, true, true, EPI, 7, 5>(p, st);
    } else if (pipe == 8) {
        if (tme == 4) return launch<256, 256, 2, 4, 64, 2, true, true, EPI, 8, 4>(p, st);
        if (tme == 5) return launch<256, 256, 2, 4, 64, 2, true, true, EPI, 8, 5>(p, st);
    }
    return NV_ERR_SHAPE;
}

template <bool A_KMAJ, bool B_KMAJ, int EPI>
int dispatch_tile(const GemmArgs& p, int tile_cfg, hipStream_t st) {
    // tile_cfg: 0 = auto (128x128 or the 256-wide tile with the planned number of fragment rows), 1 = 128x128x64 2-stage (4 waves),
    //           8 = 256x256x64 (8 waves, hand-interleaved VALU-free main loop), 84..88 = the same with TME = 4..8 fragment rows per
    //           wave, i.e. a 128 / 160 / 192 / 224 / 256 x 256 tile (88 == 8; where no cut-off instance exists the full tile runs)
    const bool can_split = p.slabs && p.counters;
    if (tile_cfg == 0) {
        // round 3: by estimated duration (est_us_*): the cut-off tiles move the break-even towards the 256-wide kernel
        double t256;
        const int tme = tme_instances<A_KMAJ, B_KMAJ, EPI> ? plan_tme(p.M, p.N, p.K, can_split, B_KMAJ, &t256)
                                                           : (t256 = est_us_256(p.M, p.N, p.K, 8, can_split, B_KMAJ), 8);
        if (est_us_128(p.M, p.N, p.K, B_KMAJ) < t256)
            return launch<128, 128, 2, 2, 64, 2, A_KMAJ, B_KMAJ, EPI>(p, st);
        return launch_tme<A_KMAJ, B_KMAJ, EPI>(p, tme, st);
    }
    if (tile_cfg == 1) return launch<128, 128, 2, 2, 64, 2, A_KMAJ, B_KMAJ, EPI>(p, st);
    if (tile_cfg == 8) return launch_tme<A_KMAJ, B_KMAJ, EPI>(p, 8, st);
    if (tile_cfg >= 84 && tile_cfg <= 88) return launch_tme<A_KMAJ, B_KMAJ, EPI>(p, tile_cfg - 80, st);
    if (tile_cfg == 94 || tile_cfg == 95) return launch_tme<A_KMAJ, B_KMAJ, EPI>(p, tile_cfg - 90, st, true);   // two-stage loop (measurement)
    return NV_ERR_ARG;
}

template <bool A_KMAJ, bool B_KMAJ>
int dispatch_epi(const GemmArgs& p, int epi, int tile_cfg, hipStream_t st) {
    switch (epi) {
        case EPI_STORE: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_STORE>(p, tile_cfg, st);
        case EPI_ACCUM: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_ACCUM>(p, tile_cfg, st);
        case EPI_RESID: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_RESID>(p, tile_cfg, st);
        case EPI_BIAS: return dispatch_tile<A_KMAJ, B_KMAJ, EPI_BIAS>(p, tile_cfg, st);
        case EPI_ROPE:            // only the packed q|k|v projection uses it: NT layout, production tiles, staged epilogue
            if constexpr (A_KMAJ && B_KMAJ) {
                return dispatch_tile<A_KMAJ, B_KMAJ, EPI_ROPE>(p, tile_cfg, st);
            }
            return NV_ERR_ARG;
        case EPI_SWIGLU_BWD:      // only the down-proj dgrad uses it: NN layout, production tiles
            if constexpr (A_KMAJ && !B_KMAJ) {
                return dispatch_tile<A_KMAJ, B_KMAJ, EPI_SWIGLU_BWD>(p, tile_cfg, st);
            }
            return NV_ERR_ARG;
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
extern "C" size_t nv_gemm_bf16_workspace_bytes() { return (size_t)MAX_SLABS * 256 * 256 * sizeof(float) + 4096; }

// workspace: nv_gemm_bf16_workspace_bytes() bytes, ZERO-FILLED once by the caller (the kernel leaves its
// ticket words zero again); NULL disables the split-K tail.
static int gemm_entry(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb,
                      int ldc, int ldr, int epilogue, int tile_cfg, void* workspace, void* stream, const void* rope_sin, int rope_S,
                      int rope_cols, const int* rope_pos = nullptr) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    if ((epilogue == EPI_RESID || epilogue == EPI_BIAS || epilogue == EPI_SWIGLU_BWD || epilogue == EPI_ROPE) && !R) return NV_ERR_ARG;
    if ((lda & 7) || (ldb & 7)) return NV_ERR_SHAPE;                 // 16-B aligned rows for the DMA
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return NV_ERR_SHAPE;
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.R = (const bf16_t*)R;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.counters = (unsigned*)workspace;
    p.slabs = workspace ? (float*)((char*)workspace + 4096) : nullptr;
    p.full_blocks = 0; p.rem = 1; p.split = 1;
    p.rope_sin = (const bf16_t*)rope_sin; p.rope_S = rope_S; p.rope_cols = rope_cols; p.rope_pos = rope_pos;
    p.b_scales = nullptr; p.fp8_epi = 0;
    {
        // tuning / measurement knobs, read once per process
        static const int env_debug = [] { const char* e = getenv("NV_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
        static const int env_group = [] { const char* e = getenv("NV_GEMM_GROUP_M"); return e ? atoi(e) : 4; }();   // 4 x 8 patch per XCD
        static const int env_order = [] { const char* e = getenv("NV_GEMM_ORDER"); return e ? atoi(e) : -1; }();    // 0 rows, 1 column strips
        static const int env_persist = [] { const char* e = getenv("NV_GEMM_PERSIST"); return e ? atoi(e) : 0; }();
        p.debug = env_debug;
        p.persist = env_persist;
        static const int env_band = [] { const char* e = getenv("NV_GEMM_BAND_REDUCE"); return e ? atoi(e) : 1; }();
        p.band_reduce = env_band;
        static const int env_pre = [] { const char* e = getenv("NV_GEMM_EPI_PRELOAD"); return e ? atoi(e) : 1; }();
        p.epi_preload = env_pre;
        static const int env_cst = [] { const char* e = getenv("NV_GEMM_C_STORE"); return e ? atoi(e) : 1; }();   // nt: +0.3 % on the step (ABAB: 43.40 / 43.50 / 43.41 / 43.57)
        p.c_nt = env_cst;
        p.group_m = env_group < 1 ? 1 : env_group;
        // default: partition the LARGER operand across the 8 XCD L2s (read once), replicate the smaller one
        p.col_strips = env_order >= 0 ? env_order : ((long)N > (long)M ? 1 : 0);
    }
    hipStream_t st = (hipStream_t)stream;
    switch (layout) {
        case 0:  // NT: A[M,K], B[N,K]
            if (K % 64) return NV_ERR_SHAPE;
            p.a_bytes = span_bytes(M, K, lda); p.b_bytes = span_bytes(N, K, ldb);
            return dispatch_epi<true, true>(p, epilogue, tile_cfg, st);
        case 1:  // NN: A[M,K], B[K,N]
            if (K % 64) return NV_ERR_SHAPE;
            p.a_bytes = span_bytes(M, K, lda); p.b_bytes = span_bytes(K, N, ldb);
            return dispatch_epi<true, false>(p, epilogue, tile_cfg, st);
        case 2:  // TN: A[K,M], B[K,N]   (K = contraction, any size: OOB k-rows read as zero)
            p.a_bytes = span_bytes(K, M, lda); p.b_bytes = span_bytes(K, N, ldb);
            return dispatch_epi<false, false>(p, epilogue, tile_cfg, st);
    }
    return NV_ERR_ARG;
}

// y = x W^T with W given as weight-only fp8: OCP e4m3fn codes [N, K] (ldq bytes per row) + one fp32 scale per output channel.
// Serves the few-hundred-row GEMMs of K/V-reuse inference steps (the shapes whose launch plan is a 128 / 160-row cut-off tile);
// any other shape returns NV_ERR_SHAPE and the caller runs nv_fp8_dequant_rows + nv_gemm_bf16 (the pre-pass form) -- a documented
// two-kernel path, not a fallback to another backend.  mode: 0 = default (nv_gemm_fp8w_default_mode / NV_GEMM_FP8_MODE, else 7), 7 = operands bf16(s*q)
// bit-exact (what the pre-pass writes), 8 = v_cvt_scalef32 with the scale as its operand, 9 = v_cvt_scalef32 unscaled + s[n] on
// the fp32 accumulator.  tile_cfg: 0 = planned, 84 / 85 = force the 128 / 160-row tile (tests).  epilogue: EPI_STORE | EPI_RESID.
// (mode 8 is a MEASUREMENT form whose results are wrong by up to 2x -- the hardware uses only the exponent of the scale operand -- so it
// can be asked for per call, never become a default: ADVICE r4)
static int g_fp8_default_mode = [] { const char* e = getenv("NV_GEMM_FP8_MODE"); const int m = e ? atoi(e) : 0; return (m == 7 || m == 9) ? m : 7; }();
// process-wide default of nv_gemm_fp8w's `mode` (what callers that pass 0 get): 7 | 9; returns the previous one; any other value only
// queries.  A deployment choice like the NV_GEMM_* environment knobs; a model's own choice travels per object instead
// (Fp8DecoderWeights.gemm_mode per call, nv_decoder_set_fp8_gemm_mode for the native layer loop), so two models with different modes
// in one process do not override each other.
extern "C" int nv_gemm_fp8w_default_mode(int mode) {
    const int prev = g_fp8_default_mode;
    if (mode == 7 || mode == 9) g_fp8_default_mode = mode;
    return prev;
}

extern "C" int nv_gemm_fp8w(const void* A, const void* codes, const float* scales, void* C, const void* R, int M, int N, int K, int lda,
                            int ldq, int ldc, int ldr, int epilogue, int mode, int tile_cfg, void* workspace, void* stream) {
    if (!A || !codes || !scales || !C || M < 0 || N < 0 || K <= 0) return NV_ERR_ARG;
    if (epilogue != EPI_STORE && epilogue != EPI_RESID) return NV_ERR_ARG;
    if (epilogue == EPI_RESID && !R) return NV_ERR_ARG;
    if (M == 0 || N == 0) return NV_OK;
    if ((lda & 7) || (ldq & 15) || (K & 63) || ((((uintptr_t)A) | ((uintptr_t)codes)) & 15) || (((uintptr_t)scales) & 15)) return NV_ERR_SHAPE;
    if (mode == 0) mode = g_fp8_default_mode;
    if (mode < 7 || mode > 9) return NV_ERR_ARG;
    const bool can_split = workspace != nullptr;
    int tme;
    if (tile_cfg == 84 || tile_cfg == 85) tme = tile_cfg - 80;
    else if (tile_cfg == 0) {
        // this kernel (128- or 160-row tile, whichever the launch model prefers) against the alternative the caller would run: the
        // de-quantisation pre-pass (3 B per weight at ~5 TB/s) + the bf16 GEMM on ITS best tile.  Relative K-step cost against the bf16
        // loop on the same tile, measured on MI355X (tools/gemm_fp8_probe.py, profiles/r04_gemm_fp8_probe.txt): mode 7 (three VALU ops per
        // pair of weights) 0.86-1.12, modes 8 / 9 (one) 0.75-0.87 -- the halved B-tile DMA and fragment reads outweigh the conversions.
        const double t4 = est_us_256(M, N, K, 4, can_split, true), t5 = est_us_256(M, N, K, 5, can_split, true);
        tme = t4 <= t5 ? 4 : 5;
        const double t_fp8 = (t4 <= t5 ? t4 : t5) * (mode == 7 ? 1.0 : 0.82);
        double t256;
        plan_tme(M, N, K, can_split, true, &t256);
        const double t128 = est_us_128(M, N, K, true);
        const double t_alt = (t128 < t256 ? t128 : t256) + (double)N * (double)K * 3.0 / 5.0e6;
        if (t_fp8 > t_alt) return NV_ERR_SHAPE;
    } else return NV_ERR_ARG;
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)codes; p.C = (bf16_t*)C; p.R = (const bf16_t*)R;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldq; p.ldc = ldc; p.ldr = ldr;
    p.counters = (unsigned*)workspace;
    p.slabs = workspace ? (float*)((char*)workspace + 4096) : nullptr;
    p.full_blocks = 0; p.rem = 1; p.split = 1;
    p.rope_sin = nullptr; p.rope_S = 1; p.rope_cols = 0; p.rope_pos = nullptr;
    p.debug = 0; p.persist = 0; p.band_reduce = 1; p.c_nt = 1; p.group_m = 4; p.epi_preload = 1;
    p.col_strips = (long)N > (long)M ? 1 : 0;
    p.b_scales = scales; p.fp8_epi = mode == 9 ? 1 : 0;
    p.a_bytes = span_bytes(M, K, lda);
    {
        const long b = (long)(N - 1) * ldq + K;                      // bytes of codes addressable from the base
        p.b_bytes = b > 0xffffffffL ? 0xffffffffu : (uint32_t)b;
    }
    hipStream_t st = (hipStream_t)stream;
    const int pipe = mode == 7 ? 7 : 8;
    return epilogue == EPI_STORE ? launch_fp8<EPI_STORE>(p, tme, pipe, st) : launch_fp8<EPI_RESID>(p, tme, pipe, st);
}

extern "C" int nv_gemm_bf16_ws(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                               int lda, int ldb, int ldc, int ldr, int epilogue, int tile_cfg, void* workspace,
                               void* stream) {
    if (epilogue == EPI_ROPE) return NV_ERR_ARG;             // has its own entry point (needs the tables)
    return gemm_entry(layout, A, B, C, R, M, N, K, lda, ldb, ldc, ldr, epilogue, tile_cfg, workspace, stream, nullptr, 1, 0);
}

// y = x W^T for the packed q|k|v projection with RoPE applied to the first `rope_cols` columns (q and k) in the epilogue:
// row m has position pos[m] (pos != NULL: packed rows) or m % S; cos/sin = nv_rope_bf16's [maxS][128] bf16 tables.  Bit-identical to
// nv_gemm_bf16(NT) followed by nv_rope_bf16; saves one read+write pass over q and k.
extern "C" int nv_gemm_bf16_rope_cfg(const void* A, const void* W, void* C, const void* rope_cos, const void* rope_sin, const int* pos,
                                     int M, int N, int K, int lda, int ldw, int ldc, int S, int rope_cols, int tile_cfg, void* workspace,
                                     void* stream) {
    if (!rope_cos || !rope_sin || (!pos && S <= 0) || rope_cols < 0 || rope_cols > N || (rope_cols & 127)) return NV_ERR_ARG;
    if ((ldc & 7) || (N & 7) || (((uintptr_t)C) & 15) || ((((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15)) return NV_ERR_SHAPE;
    return gemm_entry(0, A, W, C, rope_cos, M, N, K, lda, ldw, ldc, 0, EPI_ROPE, tile_cfg, workspace, stream, rope_sin, S > 0 ? S : 1,
                      rope_cols, pos);
}
extern "C" int nv_gemm_bf16_rope(const void* A, const void* W, void* C, const void* rope_cos, const void* rope_sin, const int* pos,
                                 int M, int N, int K, int lda, int ldw, int ldc, int S, int rope_cols, void* workspace,
                                 void* stream) {
    return nv_gemm_bf16_rope_cfg(A, W, C, rope_cos, rope_sin, pos, M, N, K, lda, ldw, ldc, S, rope_cols, 0, workspace, stream);
}

extern "C" int nv_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K,
                            int lda, int ldb, int ldc, int ldr, int epilogue, int tile_cfg, void* stream) {
    return nv_gemm_bf16_ws(layout, A, B, C, R, M, N, K, lda, ldb, ldc, ldr, epilogue, tile_cfg, nullptr, stream);
}
