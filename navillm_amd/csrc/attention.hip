// Causal + left-pad flash attention, forward and backward, head_dim = 128, bf16 in/out,
// fp32 softmax (K7b in SURVEY.md §2.2; reference: HF LlamaAttention reached from
// models/modified_lm.py:112-116, eager path = softmax(QK^T*scale + mask) @ V).
//
// Data: packed post-RoPE qkv [B*S, 3*H*128] (q | k | v), out [B*S, H*128], positions s=0..S-1
// per sample, keys s < kv_start[b] are left padding (masked), causal key <= query.
//
// CDNA4 mapping (all three kernels share it):
//  * v_mfma_f32_16x16x32_bf16 with the operands SWAPPED so the score tile comes out
//    transposed (keys down the 4 in-lane rows, one query per lane&15): the row max/sum of a
//    query is then 16 in-lane values + two cross-lane steps, the softmax rescale is a per-lane
//    scalar, and P^T is already laid out as the next MFMA's B operand (k = keys, in a
//    permuted order that the V^T operand reproduces) -- no LDS round trip for P.
//  * K/V (or Q/dO) tiles are DMA'd HBM->LDS with bounds-checked buffer_load...lds into an
//    XOR-swizzled [rows][128] image that serves both ds_read_b128 (row-major operand) and
//    ds_read_b64_tr_b16 (transposed operand) conflict-free (see key4; verified with SQ_LDS_BANK_CONFLICT); double buffered.
//  * lse is kept in the log2 domain (lse2 = m2 + log2(l)); fully masked rows store +inf so
//    the backward's exp2(s2 - lse2) is exactly 0 for them.
#include "nv_common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 128;
constexpr int ROWB = HD * 2;  // 256-B LDS rows

// XOR key of a row's sixteen 16-byte slots.  A wave's ds_read_b128 is serviced in four groups of 16 lanes that are NOT the four
// quarters of the wave -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- so the rows a
// group touches in a row-major fragment read are {0-3, 12-15} with k-group g and {4-11} with k-group g ^ 1.  The key sends the
// first set onto slots 0-7 and the second onto 8-15 (both closed under ^ 1): 16 distinct slots per group, no bank conflict.  Its upper
// three bits are distinct over any 8 aligned rows, which is what the transposing ds_read_b64_tr_b16 (half-waves, 8 rows x 32 B) needs.
// (Rounds 1-2 used ((row & 7) << 1) | ((row >> 3) & 1), built for quarter-wave groups: every b128 fragment read took 8 LDS cycles
// instead of 4 -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.31-0.40 in profiles/r03_sq_counters_prefix_episode.txt.)
__device__ __forceinline__ int key4(int row) { return (((row & 7) ^ ((row & 8) >> 1)) << 1) | ((row >> 3) & 1); }

// DMA `ROWS` rows x 128 bf16 (global row stride ld elements) into an LDS image.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_rows(const u32x4& rsrc, LDS_PTR(char) lds_p, int row0, int col0, int ld,
                                           int tid) {
    const uint32_t lds = lds_addr_of(lds_p);
    constexpr int ITERS = (ROWS * ROWB) / (NT * 16);
    static_assert((ROWS * ROWB) % (NT * 16) == 0, "tile/threads mismatch");
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int chunk = it * (NT / 64) + wave;      // 1 KiB = 4 rows
        const int row = chunk * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ key4(row);
        const uint32_t voff = (uint32_t)(((long)(row0 + row) * ld + col0 + slot * 8) * 2);
        dma16(rsrc, __builtin_amdgcn_readfirstlane(lds + chunk * 1024), voff);
    }
}

// row-major operand fragment: lane (idx=lane&15, kg=lane>>4) <- tile[r0+idx][kk*32+kg*8 .. +7]
__device__ __forceinline__ bf16x8 frag_rm(LDS_PTR(char) tile, int r0, int kk, int lane) {
    const int row = r0 + (lane & 15);
    const int slot = (kk * 4 + (lane >> 4)) ^ key4(row);
    return *(LDS_PTR(bf16x8))(tile + row * ROWB + slot * 16);
}

// transposed operand fragment: lane (idx, kg) <- { tile[rA+kg*4+e][c0+idx] , tile[rB+kg*4+e][c0+idx] } e=0..3
__device__ __forceinline__ bf16x8 frag_tr(LDS_PTR(char) tile, int rA, int rB, int c0, int lane) {
    const int t = lane & 15, kg = lane >> 4;
    const int slot = (c0 >> 3) + ((t & 3) >> 1);
    const int half = (t & 1) * 8;
    const int ra = rA + kg * 4 + (t >> 2), rb = rB + kg * 4 + (t >> 2);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + ra * ROWB + ((slot ^ key4(ra)) << 4) + half));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + rb * ROWB + ((slot ^ key4(rb)) << 4) + half));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
    u32x4 v = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}

// "this register holds its final value HERE": hipcc keeps a vector-memory load pending until the value's first use and cannot see
// the asm-issued LDS-DMA prefetches, so a fragment loaded before a tile loop and first used inside it made the loop body open with
// `s_waitcnt vmcnt(0)` -- after the next tile's prefetch had just been issued: every tile step drained the prefetch pipeline
// (round 3: SQ_WAIT_ANY 43-47 % of the wave cycles of all five attention kernels).  Pinning the loop-invariant fragments in front
// of the loop moves that wait to where it costs nothing.
template <typename V>
__device__ __forceinline__ void pin(V& v) { asm volatile("" : "+v"(v)); }

// dK/dV kernels: a 4-slot ring of (Q, dO) tiles, up to three tile prefetches (4 DMA instructions per thread each) in flight
constexpr int DKV_SLOTS = 4;
__device__ __forceinline__ void wait_stages(int still_in_flight) {
    if (still_in_flight >= 2) wait_vmcnt<8>(); else if (still_in_flight == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
}

// 2^x as the bare v_exp_f32: exp2f() wraps it in a denormal-range fix-up (compare, select, add, ldexp: 7 VALU ops per
// element against 1).  Softmax probabilities below 2^-126 are zero for every consumer here (they are rounded to bf16).
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// reduce over the 4 lane groups that share lane&15
__device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// store one gradient row fragment set (lane: 4 consecutive d per 16-wide tile dt, d = dt*16 + g*4 + r) as bf16, optionally
// through the transpose of RoPE.  Bit-identical to "round to bf16, then nv_rope_bf16(backward=1)": the partner of
// d < 64 is d + 64 = tile dt + 4 of the SAME lane.
__device__ __forceinline__ void store_grad_row(bf16_t* dst, const f32x4 (&v)[8], float scale, const bf16_t* cos_row,
                                               const bf16_t* sin_row) {
    if (!cos_row) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            u32x2 w = {pack2bf(v[dt][0] * scale, v[dt][1] * scale), pack2bf(v[dt][2] * scale, v[dt][3] * scale)};
            *(u32x2*)(dst + dt * 16) = w;
        }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const u32x2 cw = *(const u32x2*)(cos_row + dt * 16), sw = *(const u32x2*)(sin_row + dt * 16);
        const float c[4] = {__uint_as_float(cw[0] << 16), __uint_as_float(cw[0] & 0xffff0000u), __uint_as_float(cw[1] << 16),
                            __uint_as_float(cw[1] & 0xffff0000u)};
        const float sn[4] = {__uint_as_float(sw[0] << 16), __uint_as_float(sw[0] & 0xffff0000u), __uint_as_float(sw[1] << 16),
                             __uint_as_float(sw[1] & 0xffff0000u)};
        float o1[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = rbf(v[dt][r] * scale), x2 = rbf(v[dt + 4][r] * scale);
            o1[r] = rbf(x1 * c[r]) + rbf(x2 * sn[r]);
            o2[r] = rbf(x2 * c[r]) + rbf(-x1 * sn[r]);
        }
        *(u32x2*)(dst + dt * 16) = u32x2{pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3])};
        *(u32x2*)(dst + (dt + 4) * 16) = u32x2{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3])};
    }
}

struct Seq { long row0; int S, kvs, pos0, qmin; };

struct AttnArgs {
    const bf16_t* qkv; bf16_t* out; float* lse2;        // fwd
    const bf16_t* dout; const float* dsum; bf16_t* dqkv; // bwd
    const int* kv_start;
    const int* cu;         // packed ("varlen") rows: sample b = rows [cu[b], cu[b+1]); nullptr = padded [B, S] layout.  Packed
                           // samples have no padding keys: kv_start[b] is then only the POSITION of the sample's first token
                           // (the reference numbers positions over its left padding), used by the fused RoPE^T
    int B, S, H, ld;       // ld = 3*H*HD (qkv row stride), out row stride = H*HD
    int Sst;               // rows between consecutive samples in qkv/out/lse2 (= S, or a KV cache's capacity >= S)
    int q_row_min;         // only queries >= q_row_min are computed / differentiated (multiple of 128; 0 = all)
    float scale2;          // head_dim^-0.5 * log2(e)
    float scale;           // head_dim^-0.5
    const bf16_t* rope_cos; const bf16_t* rope_sin;   // backward only, optional: apply RoPE^T to dQ/dK as they are written
    float* kvacc;          // backward over a K/V cache, optional (navillm_amd/episode.py): fp32 [rows, 2*H*HD] accumulator of the gradients the
    const int* kvacc_len;  // steps of an episode send into the cached PREFIX rows: key rows < kvacc_len[b] of sample b add (kvacc_first:
    int kvacc_first;       // store) their dK | dV there straight from the fp32 MFMA accumulators and write no bf16 row
    int stat_cap;          // backward dK/dV: > 0 = the sample's lse / dsum rows are copied to LDS once per block (stat_cap floats each, a
                           // multiple of 32 >= S) and read from there; 0 = loaded from HBM inside the tile loop (S too long for the LDS budget)
    const int* dyn;        // forward over a K/V cache, optional: {S, q_row_min} read from DEVICE memory (a decode step replayed from a
                           // hipGraph: the launch arguments are frozen, the lengths are not); the grid then covers the capacity
};

// where sample b lives, how long it is, which keys are padding, its first position, and the first query row computed
// (q_row_min >= 0: that row for every sample; -1: each sample's own last 128-row block -- the pruned last layer)
__device__ __forceinline__ Seq seq_of(const AttnArgs& p, int b) {
    Seq s;
    if (p.cu) {
        const int r0 = p.cu[b];
        s.row0 = r0; s.S = p.cu[b + 1] - r0; s.kvs = 0; s.pos0 = p.kv_start[b];
    } else {
        s.row0 = (long)b * p.Sst; s.S = p.dyn ? p.dyn[0] : p.S; s.kvs = p.kv_start[b]; s.pos0 = 0;
    }
    s.qmin = p.dyn ? p.dyn[1] : p.q_row_min >= 0 ? p.q_row_min : ((s.S - 1) / 128) * 128;
    return s;
}
__device__ __forceinline__ int lse_stride(const AttnArgs& p) { return p.cu ? p.S : p.Sst; }


// =========================================================================== decode (one query row per sample)
// A decode step attends ONE new query per sample to its cached keys: the tile kernel below would push a whole 128-query block
// through the MFMAs to use one row of it (32.8 us per layer at B=8, ~650 cached tokens).  This is the HBM-bound form: block =
// (head, sample), 16 waves; a wave instruction fetches the K (then V) head slices of 4 cache rows -- 4 x 256 B, whole lines --
// lane (ks = lane>>4, dp = lane&15) holding 8 of the 128 dims of key ks.  q.k is 8 FMAs + a 16-lane butterfly, the softmax is
// online per lane group (fp32 scores and probabilities), p.V 8 FMAs; the 64 groups' (m, l, acc) meet in LDS.  K and V of the
// cache are read exactly once: 2 * L * 256 B per (sample, head).
constexpr int DEC_WAVES = 16, DEC_U = 4;
__global__ __launch_bounds__(DEC_WAVES * 64) void attn_decode_kernel(const bf16_t* __restrict__ kv, const int* __restrict__ crow,
                                                                    const int* __restrict__ pos, bf16_t* __restrict__ out, int H, int cap,
                                                                    float scale2) {
    __shared__ float pm[DEC_WAVES * 4], pl[DEC_WAVES * 4];
    __shared__ float pacc[DEC_WAVES * 4][HD + 1];
    const int h = blockIdx.x, r = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = lane >> 4, dp = lane & 15;
    const int ld = 3 * H * HD;
    const int L = min(pos[r] + 1, cap);                            // keys 0 .. pos (causal, left-aligned cache)
    const bf16_t* base = kv + (long)r * cap * ld;
    float q[8];
    {
        const u32x4 v = *(const u32x4*)(kv + (long)crow[r] * ld + h * HD + dp * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { q[2 * j] = __uint_as_float(v[j] << 16) * scale2; q[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u) * scale2; }
    }
    const int kcol = H * HD + h * HD + dp * 8, vcol = 2 * H * HD + h * HD + dp * 8;
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = wave * 4 + ks; k0 < L; k0 += 64 * DEC_U) {           // a 16-lane group shares k0: its lanes leave the loop together
        u32x4 kf[DEC_U], vf[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int key = k0 + u * 64;
            const long row = key < L ? key : L - 1;
            kf[u] = __builtin_nontemporal_load((const u32x4*)(base + row * ld + kcol));
            vf[u] = __builtin_nontemporal_load((const u32x4*)(base + row * ld + vcol));
        }
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const int key = k0 + u * 64;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s += q[2 * j] * __uint_as_float(kf[u][j] << 16);
                s += q[2 * j + 1] * __uint_as_float(kf[u][j] & 0xffff0000u);
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            if (key < L) {                                          // uniform within the 16-lane group
                const float mn = fmaxf(m, s);
                const float sc = fast_exp2(m - mn), p = fast_exp2(s - mn);   // m = -inf the first time: exp2(-inf) = 0
                l = l * sc + p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] = acc[2 * j] * sc + p * __uint_as_float(vf[u][j] << 16);
                    acc[2 * j + 1] = acc[2 * j + 1] * sc + p * __uint_as_float(vf[u][j] & 0xffff0000u);
                }
                m = mn;
            }
        }
    }
    const int g = wave * 4 + ks;
    if (dp == 0) { pm[g] = m; pl[g] = l; }
#pragma unroll
    for (int j = 0; j < 8; ++j) pacc[g][dp * 8 + j] = acc[j];
    __syncthreads();
    if (tid < HD) {
        float mm = -INFINITY;
        for (int i = 0; i < DEC_WAVES * 4; ++i) mm = fmaxf(mm, pm[i]);
        float num = 0.f, den = 0.f;
        for (int i = 0; i < DEC_WAVES * 4; ++i) {
            if (pm[i] != -INFINITY) {
                const float w = fast_exp2(pm[i] - mm);
                num += pacc[i][tid] * w;
                den += pl[i] * w;
            }
        }
        out[((long)r * H + h) * HD + tid] = f2bf(den > 0.f ? num / den : 0.f);
    }
}

// =========================================================================== forward
// grid (ceil(S/128), B*H), 256 threads: wave w owns queries q0 + w*32 .. +31 (two 16-query tiles)
//
// HFR = false: the product kernel (flash style: fp32 scores, unnormalised bf16 P, one pass).
// HFR = true : PARITY INSTRUMENT, not on the product path.  Reproduces the rounding points of HF's eager LlamaAttention
// in bf16 (transformers modeling_llama.py eager_attention_forward, reached from models/modified_lm.py:112-116):
//   scores = bf16(q k^T) ; scores = bf16(scores * hd^-0.5) ; P = bf16(softmax_fp32(scores)) ; out = bf16(P v)
// which needs the row's final max and sum before P can be rounded -> two passes over the key tiles (pass 0: statistics,
// pass 1: normalised bf16 P and the PV product).  tests/test_round2_gpu.py uses it to show that the ~1e-2 distance
// between the product kernel and the reference's bf16 run is exactly this difference in rounding points.
template <bool HFR>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 64 * ROWB;  // 16 KiB per K or V tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid (B*H, query blocks): x walks the heads, y the query blocks from the LAST one down -- causal work grows with the
    // query index, so the long blocks are dispatched first and the short ones fill the tail of the launch
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const Seq sq = seq_of(p, b);
    const int S = sq.S, ld = p.ld;
    const int q0 = sq.qmin + (gridDim.y - 1 - blockIdx.y) * 128;
    if (q0 >= S) return;                                       // packed rows: a shorter sample has fewer query blocks
    const int kvs = sq.kvs;
    const int qi = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + sq.row0 * ld;
    const uint32_t span = (uint32_t)(((long)(S - 1) * ld + 3 * p.H * HD) * 2);
    const u32x4 rs = make_desc(base, span);
    const int kcol = p.H * HD + h * HD, vcol = 2 * p.H * HD + h * HD;

    // Q fragments (B operand: j = query, k = head dim), straight from HBM
    bf16x8 qf[2][4];
    int qpos[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qpos[j] = q0 + wave * 32 + j * 16 + qi;
        const int ql = qpos[j] < S ? qpos[j] : S - 1;
        const bf16_t* qp = base + (long)ql * ld + h * HD + g * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[j][kk] = *(const bf16x8*)(qp + kk * 32);
    }

    f32x4 o[8][2];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[dt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m2[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

    const int q_hi = (q0 + 127 < S - 1) ? q0 + 127 : S - 1;   // last query of this block
    const int kt_beg = kvs / 64, kt_end = q_hi / 64;           // inclusive key-tile range
    if (kt_beg <= kt_end) {
        auto stage = [&](int kt, int buf) {
            stage_rows<64, 256>(rs, smem + buf * 2 * TILE, kt * 64, kcol, ld, tid);
            stage_rows<64, 256>(rs, smem + buf * 2 * TILE + TILE, kt * 64, vcol, ld, tid);
        };
        float hf_inv[2] = {0.f, 0.f};
#pragma unroll 1
        for (int pass = 0; pass < (HFR ? 2 : 1); ++pass) {
        if (HFR && pass == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float lt = grp_sum(l[j]);
                hf_inv[j] = lt > 0.f ? 1.f / lt : 0.f;
            }
        }
        stage(kt_beg, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pin(qf[j][kk]);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int kt = kt_beg; kt <= kt_end; ++kt) {
            if (kt < kt_end) stage(kt + 1, cur ^ 1);
            LDS_PTR(char) sk = smem + cur * 2 * TILE;
            LDS_PTR(char) sv = sk + TILE;
            // ---- S^T = K Q^T : 4 key tiles x 2 query tiles
            f32x4 s[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 kf = frag_rm(sk, i * 16, kk, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        s[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[j][kk], s[i][j], 0, 0, 0);
                }
            }
            // ---- mask + online softmax (per query = per lane&15).  Key tiles entirely below this wave's first query and
            // entirely at/after the left padding need no mask at all (wave-uniform test): most tiles of a long prompt.
            const bool interior = !HFR && (kt * 64 + 63 <= q0 + wave * 32) && (kt * 64 >= kvs);
            if constexpr (HFR) {
                constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt * 64 + i * 16 + g * 4 + r;
                            const bool ok = (key <= qpos[j]) && (key >= kvs);
                            const float v = ok ? rbf(rbf(s[i][j][r]) * p.scale) * LOG2E : -INFINITY;   // HF's two bf16 roundings
                            s[i][j][r] = v;
                            mx = fmaxf(mx, v);
                        }
                    if (pass == 0) {
                        mx = grp_max(mx);
                        const float mn = fmaxf(m2[j], mx);
                        const float msafe = (mn == -INFINITY) ? 0.f : mn;
                        const float alpha = fast_exp2(m2[j] - msafe);
                        m2[j] = mn;
                        float rs_ = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) rs_ += fast_exp2(s[i][j][r] - msafe);
                        l[j] = l[j] * alpha + rs_;
                    } else {
                        const float msafe = (m2[j] == -INFINITY) ? 0.f : m2[j];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[i][j][r] = fast_exp2(s[i][j][r] - msafe) * hf_inv[j];   // normalised P (bf16 in pack_frag)
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float mx = -INFINITY;                          // max of the RAW scores (scale2 > 0 commutes with max)
                if (interior) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[i][j][r]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt * 64 + i * 16 + g * 4 + r;
                            const bool ok = (key <= qpos[j]) && (key >= kvs);
                            const float v = ok ? s[i][j][r] : -INFINITY;
                            s[i][j][r] = v;
                            mx = fmaxf(mx, v);
                        }
                }
                mx = grp_max(mx) * p.scale2;
                const float mn = fmaxf(m2[j], mx);
                const float msafe = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = fast_exp2(m2[j] - msafe);   // m2=-inf -> 0
                m2[j] = mn;
                float rs_ = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = fast_exp2(fmaf(s[i][j][r], p.scale2, -msafe));   // masked: fma(-inf, +c, x) = -inf -> 0
                        s[i][j][r] = e;
                        rs_ += e;
                    }
                l[j] = l[j] * alpha + rs_;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) o[dt][j] *= alpha;
            }
            }
            // ---- O^T += V^T P^T : k = keys (two 32-key steps), 8 d tiles
            if (!HFR || pass == 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                bf16x8 pf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) pf[j] = pack_frag(s[2 * a][j], s[2 * a + 1][j]);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    const bf16x8 vf = frag_tr(sv, a * 32, a * 32 + 16, dt * 16, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        o[dt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[j], o[dt][j], 0, 0, 0);
                }
            }
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        }
    }
    // ---- finalize
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lt = grp_sum(l[j]);
        const bool valid = qpos[j] < S;
        const float inv = HFR ? (lt > 0.f ? 1.f : 0.f) : (lt > 0.f ? 1.f / lt : 0.f);   // HFR: P was normalised before the PV product
        if (valid) {
            bf16_t* op = p.out + (sq.row0 + qpos[j]) * (p.H * HD) + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                u32x2 w = {pack2bf(o[dt][j][0] * inv, o[dt][j][1] * inv), pack2bf(o[dt][j][2] * inv, o[dt][j][3] * inv)};
                *(u32x2*)(op + dt * 16) = w;
            }
            if (g == 0) p.lse2[((long)b * p.H + h) * lse_stride(p) + qpos[j]] = lt > 0.f ? m2[j] + log2f(lt) : INFINITY;
        }
    }
}

// =========================================================================== episode forward (round 5)
// The forward twin of the episode backward kernels below: the steps of a prefix-reuse episode (navillm_amd/episode.py) keep their
// q|k|v rows in per-layer episode buffers -- prefix rows of sample b at [cu[b], cu[b+1]), step t's rows of sample b at
// [off[t,b], off[t,b] + n[t,b]) -- and a step's queries see their sample's whole prefix and the step's own earlier rows.  Round 4 ran
// the batched (teacher-forced) forward's attention per step through the K/V-cache layout: scatter the step's rows into the cache,
// one strided forward launch, gather the outputs back -- 18 launches per layer for six steps.  This kernel reads the row buffers in
// place, ALL steps in one launch: grid (B*H, T * query blocks); key tile kt covers the VIRTUAL keys [64 kt, 64 kt + 64) of the
// (sample, step) -- virtual key v is prefix row v for v < lp, the step's row v - lp otherwise -- i.e. exactly the cache positions
// of the round-4 form, so every query row sees the same key tiles in the same order: bit-identical outputs and lse.
struct EpiFwdArgs {
    const bf16_t* qkv; bf16_t* out;
    float* const* lse;              // [T] device pointers: lse2 [B, H, cap] of each step, written at cache position lp + j
    const int* cu;                  // [B + 1] prefix rows
    const int* tab;                 // [T * B] off | [T * B] n
    int T, B, H, ld, cap, QB;       // QB = query blocks (128 rows) per (step, sample)
    uint32_t span;                  // bytes addressable from qkv (rows of the episode buffer * ld * 2)
    float scale2;
};

__global__ __launch_bounds__(256, 2) void epi_fwd_kernel(EpiFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    // y walks the (step, query block) pairs with the LAST query block of every step first (the longest key ranges)
    const int yy = (int)(gridDim.y - 1 - blockIdx.y);
    const int t = yy % p.T, qb = yy / p.T;
    const int cub = p.cu[b], lp = p.cu[b + 1] - cub;
    const int off = p.tab[t * p.B + b], n = p.tab[p.T * p.B + t * p.B + b];
    const int q0 = qb * 128;                                   // step-local index of this block's first query
    if (q0 >= n) return;
    const int ld = p.ld;
    const int qi = lane & 15, g = lane >> 4;
    const u32x4 rs = make_desc(p.qkv, p.span);
    const int kcol = p.H * HD + h * HD, vcol = 2 * p.H * HD + h * HD;
    const int Sv = lp + n;                                     // virtual sequence length of (sample, step)

    bf16x8 qf[2][4];
    int qpos[2];                                               // VIRTUAL positions of this lane's two queries
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int jl = q0 + wave * 32 + j * 16 + qi;
        qpos[j] = lp + jl;
        const long row = off + (jl < n ? jl : n - 1);
        const bf16_t* qp = p.qkv + row * ld + h * HD + g * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[j][kk] = *(const bf16x8*)(qp + kk * 32);
    }
    f32x4 o[8][2];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[dt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m2[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

    const int q_hi = lp + ((q0 + 127 < n - 1) ? q0 + 127 : n - 1);       // last (virtual) query of this block
    const int kt_end = q_hi / 64;
    // DMA of one 64-key tile of K and of V: per lane the PHYSICAL row of its virtual key (clamped to the last valid one: such keys
    // lie behind every query of the block and are masked)
    auto stage = [&](int kt, int buf) {
        const uint32_t lk = lds_addr_of(smem + buf * 2 * TILE), lv = lk + TILE;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int chunk = it * 4 + wave;
            const int row = chunk * 4 + (lane >> 4);
            int vk = kt * 64 + row;
            vk = vk < Sv ? vk : Sv - 1;
            const long prow = vk < lp ? (long)cub + vk : (long)off + (vk - lp);
            const int slot = (lane & 15) ^ key4(row);
            const uint32_t vo = (uint32_t)((prow * ld + slot * 8) * 2);
            dma16(rs, __builtin_amdgcn_readfirstlane(lk + chunk * 1024), vo + (uint32_t)(kcol * 2));
            dma16(rs, __builtin_amdgcn_readfirstlane(lv + chunk * 1024), vo + (uint32_t)(vcol * 2));
        }
    };
    stage(0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pin(qf[j][kk]);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt <= kt_end; ++kt) {
        if (kt < kt_end) stage(kt + 1, cur ^ 1);
        LDS_PTR(char) sk = smem + cur * 2 * TILE;
        LDS_PTR(char) sv = sk + TILE;
        f32x4 s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 kf = frag_rm(sk, i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) s[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[j][kk], s[i][j], 0, 0, 0);
            }
        }
        const bool interior = kt * 64 + 63 <= lp + q0 + wave * 32;      // (same test as attn_fwd_kernel, in virtual positions)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mx = -INFINITY;
            if (interior) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[i][j][r]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + i * 16 + g * 4 + r;
                        const float v = key <= qpos[j] ? s[i][j][r] : -INFINITY;
                        s[i][j][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            mx = grp_max(mx) * p.scale2;
            const float mn = fmaxf(m2[j], mx);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = fast_exp2(m2[j] - msafe);
            m2[j] = mn;
            float rs_ = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(fmaf(s[i][j][r], p.scale2, -msafe));
                    s[i][j][r] = e;
                    rs_ += e;
                }
            l[j] = l[j] * alpha + rs_;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) o[dt][j] *= alpha;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            bf16x8 pf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) pf[j] = pack_frag(s[2 * a][j], s[2 * a + 1][j]);
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const bf16x8 vf = frag_tr(sv, a * 32, a * 32 + 16, dt * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) o[dt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[j], o[dt][j], 0, 0, 0);
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    float* lse2 = p.lse[t] + ((long)b * p.H + h) * p.cap;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lt = grp_sum(l[j]);
        const int jl = qpos[j] - lp;
        if (jl < n) {
            const float inv = lt > 0.f ? 1.f / lt : 0.f;
            bf16_t* op = p.out + ((long)off + jl) * (p.H * HD) + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                u32x2 w = {pack2bf(o[dt][j][0] * inv, o[dt][j][1] * inv), pack2bf(o[dt][j][2] * inv, o[dt][j][3] * inv)};
                *(u32x2*)(op + dt * 16) = w;
            }
            if (g == 0) lse2[qpos[j]] = lt > 0.f ? m2[j] + log2f(lt) : INFINITY;
        }
    }
}

// =========================================================================== backward prep
// dsum[b,h,q] = sum_d dO[q,d] * O[q,d]   (16 lanes per (row, head): 16-B loads, 4 items per wave)
// rows = B*S (padded layout) or cu[B] (packed layout: the sample of a row is found by walking cu, B is small)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                            float* __restrict__ dsum, const int* __restrict__ cu, long rows, int B,
                                                            int S, int H, int q_lo, int nq) {
    const int sub = threadIdx.x & 15;
    const long item = (long)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, head)
    const bool live = item < rows * H;
    long row = live ? item / H : 0;
    const int h = live ? (int)(item % H) : 0;
    int bb = 0, s_ = 0;
    if (cu) {
        while (bb + 1 < B && row >= cu[bb + 1]) ++bb;
        s_ = (int)(row - cu[bb]);
    } else {
        // padded / K/V-cache layout: only the query rows [q_lo, q_lo + nq) of every sample carry gradient (round 3: a suffix step of
        // navillm_amd/episode.py differentiates ~130 of the cache's 1024 rows per sample; the other rows' dsum is never read)
        bb = (int)(row / nq); s_ = q_lo + (int)(row % nq);
        row = (long)bb * S + s_;
    }
    const u32x4 av = *(const u32x4*)(dout + row * (H * HD) + h * HD + sub * 8);
    const u32x4 ov = *(const u32x4*)(out + row * (H * HD) + h * HD + sub * 8);
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        v += __uint_as_float(av[q] << 16) * __uint_as_float(ov[q] << 16) +
             __uint_as_float(av[q] & 0xffff0000u) * __uint_as_float(ov[q] & 0xffff0000u);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (live && sub == 0) dsum[((long)bb * H + h) * S + s_] = v;
}

// =========================================================================== backward dK, dV
// grid (ceil(S/(64*KW)), B*H), 256 threads: wave w owns KW key tiles of 16 (keys kblk + (w*KW+jk)*16 ..); walks query
// tiles of 32.  KW = 2 halves the LDS fragment traffic per MFMA (every Q/dO fragment feeds two key tiles).
template <int KW>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 32 * ROWB;  // 8 KiB per Q or dO tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;      // grid (B*H, key blocks): key block 0 (all queries) goes first
    const Seq sq = seq_of(p, b);
    const int S = sq.S, ld = p.ld;
    const int kblk = blockIdx.y * 64 * KW;
    if (kblk >= S) return;
    const int kvs = sq.kvs;
    const int ki = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + sq.row0 * ld;
    const uint32_t span = (uint32_t)(((long)(S - 1) * ld + 3 * p.H * HD) * 2);
    const u32x4 rq = make_desc(base, span);
    const int od = p.H * HD;
    const bf16_t* dobase = p.dout + sq.row0 * od;
    const u32x4 rdo = make_desc(dobase, (uint32_t)(((long)(S - 1) * od + od) * 2));
    const float* lse2 = p.lse2 + ((long)b * p.H + h) * lse_stride(p);
    const float* dsum = p.dsum + ((long)b * p.H + h) * lse_stride(p);

    // K and V fragments (B operand: j = key, k = head dim) from HBM
    bf16x8 kf[KW][4], vf[KW][4];
    int key[KW];
#pragma unroll
    for (int jk = 0; jk < KW; ++jk) {
        key[jk] = kblk + (wave * KW + jk) * 16 + ki;
        const int kl = key[jk] < S ? key[jk] : S - 1;
        const bf16_t* kp = base + (long)kl * ld + p.H * HD + h * HD + g * 8;
        const bf16_t* vp = base + (long)kl * ld + 2 * p.H * HD + h * HD + g * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[jk][kk] = *(const bf16x8*)(kp + kk * 32);
            vf[jk][kk] = *(const bf16x8*)(vp + kk * 32);
        }
    }
    f32x4 dv[KW][8], dk[KW][8];
#pragma unroll
    for (int jk = 0; jk < KW; ++jk)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) { dv[jk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[jk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // queries that can see this key block: q >= kblk (causal), q >= kvs (pad queries have p=0 anyway)
    const int qt_lo = sq.qmin / 32;
    const int qt_beg = kblk / 32 > qt_lo ? kblk / 32 : qt_lo, qt_end = (S - 1) / 32;
    auto stage = [&](int qt, int buf) {
        stage_rows<32, 256>(rq, smem + buf * 2 * TILE, qt * 32, h * HD, ld, tid);
        stage_rows<32, 256>(rdo, smem + buf * 2 * TILE + TILE, qt * 32, h * HD, od, tid);
    };
    if (qt_beg <= qt_end) {
    // The softmax statistics of every query this block will meet go to LDS FIRST.  vmcnt retires in order and hipcc does not see the
    // asm-issued DMA prefetches: a global load inside the tile loop made it wait with vmcnt(0) at the load's first use, i.e. for
    // the prefetch issued a moment earlier -- every tile step paid a full DMA round trip (round 3: SQ_WAIT_ANY 46 % of the wave
    // cycles).  With the statistics in LDS the loop has no compiler-visible vector-memory load left.
    LDS_PTR(float) s_lse = (LDS_PTR(float))(smem + DKV_SLOTS * 2 * TILE);
    LDS_PTR(float) s_ds = s_lse + p.stat_cap;
    if (p.stat_cap) {
        for (int i = qt_beg * 32 + tid; i < p.stat_cap; i += 256) {
            const int qc = i < S ? i : S - 1;
            s_lse[i] = lse2[qc];
            s_ds[i] = dsum[qc];
        }
        __syncthreads();
    }
    // 4-slot ring, three tiles in flight: a tile step here is short (32-64 MFMAs = 0.25-0.5 us), several times shorter than the DMA
    // round trip (round 1: one tile in flight; round 2: two; round 3: three, once the loop stopped draining them -- see pin())
    stage(qt_beg, 0);
    if (qt_beg < qt_end) stage(qt_beg + 1, 1);
    if (qt_beg + 1 < qt_end) stage(qt_beg + 2, 2);
#pragma unroll
    for (int jk = 0; jk < KW; ++jk)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { pin(kf[jk][kk]); pin(vf[jk][kk]); }
    wait_stages(qt_end - qt_beg);
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int qt = qt_beg; qt <= qt_end; ++qt) {
        float lq[2][4], dq_[2][4];
        if (p.stat_cap) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a4 = *(LDS_PTR(f32x4))(s_lse + qt * 32 + j * 16 + g * 4), d4 = *(LDS_PTR(f32x4))(s_ds + qt * 32 + j * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { lq[j][r] = a4[r]; dq_[j][r] = d4[r]; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * 32 + j * 16 + g * 4 + r;
                    const int qc = q < S ? q : S - 1;
                    lq[j][r] = lse2[qc];
                    dq_[j][r] = dsum[qc];
                }
        }
        if (qt + 3 <= qt_end) stage(qt + 3, (cur + 3) & 3);
        LDS_PTR(char) sq = smem + cur * 2 * TILE;
        LDS_PTR(char) sdo = sq + TILE;
        // ---- S = Q K^T and dP = dO V^T : lane holds key = lane&15, queries qt*32 + j*16 + g*4 + r
        f32x4 s[KW][2], dp[KW][2];
#pragma unroll
        for (int jk = 0; jk < KW; ++jk)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[jk][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[jk][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // reads of two k-steps back to back, then their MFMAs (see epi_bwd_dkv_kernel)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            bf16x8 qa[2][2], da[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    qa[kk][j] = frag_rm(sq, j * 16, k2 * 2 + kk, lane);
                    da[kk][j] = frag_rm(sdo, j * 16, k2 * 2 + kk, lane);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int jk = 0; jk < KW; ++jk) {
                        s[jk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[kk][j], kf[jk][k2 * 2 + kk], s[jk][j], 0, 0, 0);
                        dp[jk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[kk][j], vf[jk][k2 * 2 + kk], dp[jk][j], 0, 0, 0);
                    }
        }
        bf16x8 pfrag[KW], dsfrag[KW];
        // query tiles entirely past this wave's keys, inside [0, S), with the keys past the left padding: no mask
        const int wkey_lo = kblk + wave * KW * 16, wkey_hi = wkey_lo + KW * 16 - 1;
        const bool interior = (qt * 32 >= wkey_hi) && (qt * 32 + 31 < S) && (wkey_lo >= kvs) && (wkey_hi < S);
#pragma unroll
        for (int jk = 0; jk < KW; ++jk) {
            f32x4 pv[2], ds[2];
            if (interior) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = fast_exp2(fmaf(s[jk][j][r], p.scale2, -lq[j][r]));
                        pv[j][r] = pe;
                        ds[j][r] = pe * (dp[jk][j][r] - dq_[j][r]);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = qt * 32 + j * 16 + g * 4 + r;
                        const bool ok = (q < S) && (key[jk] <= q) && (key[jk] >= kvs) && (key[jk] < S);
                        const float pe = ok ? fast_exp2(fmaf(s[jk][j][r], p.scale2, -lq[j][r])) : 0.f;
                        pv[j][r] = pe;
                        ds[j][r] = pe * (dp[jk][j][r] - dq_[j][r]);
                    }
            }
            pfrag[jk] = pack_frag(pv[0], pv[1]);
            dsfrag[jk] = pack_frag(ds[0], ds[1]);
        }
        // ---- dV^T += dO^T P ; dK^T += Q^T dS   (k = the 32 queries, permuted as pack_frag lays them)
#pragma unroll
        for (int d4 = 0; d4 < 2; ++d4) {
            bf16x8 dot_[4], qt_[4];
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                dot_[dd] = frag_tr(sdo, 0, 16, (d4 * 4 + dd) * 16, lane);
                qt_[dd] = frag_tr(sq, 0, 16, (d4 * 4 + dd) * 16, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                for (int jk = 0; jk < KW; ++jk) {
                    dv[jk][d4 * 4 + dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_[dd], pfrag[jk], dv[jk][d4 * 4 + dd], 0, 0, 0);
                    dk[jk][d4 * 4 + dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_[dd], dsfrag[jk], dk[jk][d4 * 4 + dd], 0, 0, 0);
                }
        }
        wait_stages(qt_end - (qt + 1));                                     // tile qt+1 has landed (qt+2, qt+3 may be in flight)
        __builtin_amdgcn_s_barrier();
        cur = (cur + 1) & 3;
    }
    }
    const int acc_len = p.kvacc ? p.kvacc_len[b] : 0;
#pragma unroll
    for (int jk = 0; jk < KW; ++jk)
        if (key[jk] < S) {
            if (key[jk] < acc_len) {
                // a cached prefix row: its K/V gradient is summed over the episode's steps in fp32 (what nv_kv_grad_accum_f32 did from
                // the bf16 row in a second pass: 342 MB per layer and step at 7B/B=8); same scaling as the bf16 store, no rounding
                float* ak = p.kvacc + (sq.row0 + key[jk]) * (2L * p.H * HD) + h * HD + g * 4;
                float* av = ak + p.H * HD;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    f32x4 k4 = dk[jk][dt] * p.scale, v4 = dv[jk][dt];
                    if (!p.kvacc_first) { k4 += *(const f32x4*)(ak + dt * 16); v4 += *(const f32x4*)(av + dt * 16); }
                    *(f32x4*)(ak + dt * 16) = k4;
                    *(f32x4*)(av + dt * 16) = v4;
                }
                continue;
            }
            bf16_t* kp = p.dqkv + (sq.row0 + key[jk]) * ld + p.H * HD + h * HD + g * 4;
            bf16_t* vp = p.dqkv + (sq.row0 + key[jk]) * ld + 2 * p.H * HD + h * HD + g * 4;
            store_grad_row(kp, dk[jk], p.scale, p.rope_cos ? p.rope_cos + (long)(key[jk] + sq.pos0) * HD + g * 4 : nullptr,
                           p.rope_sin ? p.rope_sin + (long)(key[jk] + sq.pos0) * HD + g * 4 : nullptr);
            store_grad_row(vp, dv[jk], 1.f, nullptr, nullptr);
        }
}

// =========================================================================== backward dQ
// grid (ceil((S-q_row_min)/(64*QW)), B*H), 256 threads: wave w owns QW query tiles of 16; walks key tiles of 64.
template <int QW>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;      // grid (B*H, query blocks), last (longest) query block first
    const Seq sq = seq_of(p, b);
    const int S = sq.S, ld = p.ld;
    const int q0 = sq.qmin + (gridDim.y - 1 - blockIdx.y) * 64 * QW;
    if (q0 >= S) return;
    const int kvs = sq.kvs;
    const int qi = lane & 15, g = lane >> 4;
    const bf16_t* base = p.qkv + sq.row0 * ld;
    const uint32_t span = (uint32_t)(((long)(S - 1) * ld + 3 * p.H * HD) * 2);
    const u32x4 rs = make_desc(base, span);
    const int od = p.H * HD;
    const int kcol = p.H * HD + h * HD, vcol = 2 * p.H * HD + h * HD;

    bf16x8 qf[QW][4], dof[QW][4];
    int q[QW];
    float my_lse[QW], my_ds[QW];
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        q[j] = q0 + (wave * QW + j) * 16 + qi;
        const int ql = q[j] < S ? q[j] : S - 1;
        const bf16_t* qp = base + (long)ql * ld + h * HD + g * 8;
        const bf16_t* dp_ = p.dout + (sq.row0 + ql) * od + h * HD + g * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[j][kk] = *(const bf16x8*)(qp + kk * 32);
            dof[j][kk] = *(const bf16x8*)(dp_ + kk * 32);
        }
        my_lse[j] = p.lse2[((long)b * p.H + h) * lse_stride(p) + ql];
        my_ds[j] = p.dsum[((long)b * p.H + h) * lse_stride(p) + ql];
    }
    f32x4 dq[QW][8];
#pragma unroll
    for (int j = 0; j < QW; ++j)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) dq[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q_hi = (q0 + 64 * QW - 1 < S - 1) ? q0 + 64 * QW - 1 : S - 1;
    const int kt_beg = kvs / 64, kt_end = q_hi / 64;
    if (kt_beg <= kt_end) {
        auto stage = [&](int kt, int buf) {
            stage_rows<64, 256>(rs, smem + buf * 2 * TILE, kt * 64, kcol, ld, tid);
            stage_rows<64, 256>(rs, smem + buf * 2 * TILE + TILE, kt * 64, vcol, ld, tid);
        };
        stage(kt_beg, 0);
#pragma unroll
        for (int j = 0; j < QW; ++j) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { pin(qf[j][kk]); pin(dof[j][kk]); }
            pin(my_lse[j]); pin(my_ds[j]);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int kt = kt_beg; kt <= kt_end; ++kt) {
            if (kt < kt_end) stage(kt + 1, cur ^ 1);
            LDS_PTR(char) sk = smem + cur * 2 * TILE;
            LDS_PTR(char) sv = sk + TILE;
            f32x4 s[4][QW], dp[4][QW];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < QW; ++j) { s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // one k-step's eight fragment reads back to back, then its MFMAs (see epi_bwd_dkv_kernel)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 ka[4], va[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ka[i] = frag_rm(sk, i * 16, kk, lane);
                    va[i] = frag_rm(sv, i * 16, kk, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < QW; ++j) {
                        s[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[i], qf[j][kk], s[i][j], 0, 0, 0);
                        dp[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[i], dof[j][kk], dp[i][j], 0, 0, 0);
                    }
            }
            // key tiles entirely before this wave's first query, past the left padding, with all its queries < S: no mask
            const int wq_lo = q0 + wave * QW * 16;
            const bool interior = (kt * 64 + 63 <= wq_lo) && (kt * 64 >= kvs) && (wq_lo + QW * 16 - 1 < S);
#pragma unroll
            for (int j = 0; j < QW; ++j) {
                if (interior) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pe = fast_exp2(fmaf(s[i][j][r], p.scale2, -my_lse[j]));
                            s[i][j][r] = pe * (dp[i][j][r] - my_ds[j]);   // dS^T
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt * 64 + i * 16 + g * 4 + r;
                            const bool ok = (q[j] < S) && (key <= q[j]) && (key >= kvs);
                            const float pe = ok ? fast_exp2(fmaf(s[i][j][r], p.scale2, -my_lse[j])) : 0.f;
                            s[i][j][r] = pe * (dp[i][j][r] - my_ds[j]);   // dS^T
                        }
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                bf16x8 dsf[QW];
#pragma unroll
                for (int j = 0; j < QW; ++j) dsf[j] = pack_frag(s[2 * a][j], s[2 * a + 1][j]);
                bf16x8 kt_[8];
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) kt_[dt] = frag_tr(sk, a * 32, a * 32 + 16, dt * 16, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int j = 0; j < QW; ++j) dq[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt], dsf[j], dq[j][dt], 0, 0, 0);
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int j = 0; j < QW; ++j)
        if (q[j] < S) {
            bf16_t* qp = p.dqkv + (sq.row0 + q[j]) * ld + h * HD + g * 4;
            store_grad_row(qp, dq[j], p.scale, p.rope_cos ? p.rope_cos + (long)(q[j] + sq.pos0) * HD + g * 4 : nullptr,
                           p.rope_sin ? p.rope_sin + (long)(q[j] + sq.pos0) * HD + g * 4 : nullptr);
        }
}

// =========================================================================== backward of a prefix-reuse EPISODE's steps (round 3)
// navillm_amd/episode.py, mode "all": at finish_episode() one layer's q|k|v, attention outputs and output gradients of EVERY token row
// of the episode sit in [R, .] buffers -- the prompt prefixes packed (sample b = rows [cu[b], cu[b+1])), then one block per step,
// packed as well (step t, sample b = rows off[t,b] + j, j < n[t,b]).  A step's queries see their sample's whole
// prefix and the step's own earlier rows.  One launch per kernel covers all T steps: the prefix key blocks walk the query tiles of
// every step and leave the fp32 sum of the K/V gradients in kv_acc (round 3a ran one launch per step, each re-reading and
// re-writing 139 MB of fp32 accumulator); nothing is scattered into the K/V cache layout and back.
// descriptor whose base / span come from values LOADED in the kernel (wave-uniform, but in vector registers)
__device__ __forceinline__ u32x4 make_desc_u(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}

constexpr int EPI_TMAX = 128;       // steps per episode the kernels keep a table for

struct EpiArgs {
    const bf16_t* qkv; const bf16_t* dout; bf16_t* dqkv;
    const float* dsum;              // [H, R - Mp]: row-sums of dO * O over the steps' rows
    const float* const* lse;        // [T] device pointers: lse2 [B, H, cap] of each step, indexed by cache position (prefix_len + j)
    const int* cu;                  // [B + 1]
    const int* tab;                 // [T * B] off (first row of step t, sample b) | [T * B] n (its rows)
    float* kvacc;                   // fp32 [B * cap, 2 * H * HD], row b * cap + key
    const bf16_t* rope_cos; const bf16_t* rope_sin;
    int T, B, H, ld, cap, Mp, Rs, nPB, nSB;
    int stat_cap;                   // floats per statistics row in LDS: N_max rounded up to the 32-query tile
    int kvacc_add;                  // != 0: the prefix rows' dK | dV are ADDED to kvacc (a later segment of a long episode), else stored
    float scale2, scale;
};

__global__ __launch_bounds__(256) void epi_bwd_dkv_kernel(EpiArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 32 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, B = p.B, ld = p.ld, od = p.H * HD;
    const int lp = p.cu[b + 1] - p.cu[b];
    const bool pre = (int)blockIdx.y < p.nPB;          // a block of prefix keys (queries: every step) or of one step's own keys
    int t_beg, t_end, kblk, klen;
    long krow0;
    if (pre) {
        kblk = blockIdx.y * 64; t_beg = 0; t_end = T; krow0 = p.cu[b]; klen = lp;
    } else {
        const int y = blockIdx.y - p.nPB;
        t_beg = y / p.nSB; t_end = t_beg + 1; kblk = (y % p.nSB) * 64;
        klen = p.tab[T * B + t_beg * B + b]; krow0 = p.tab[t_beg * B + b];
    }
    if (kblk >= klen) return;
    const int ki = lane & 15, g = lane >> 4;
    const int key = kblk + wave * 16 + ki;
    bf16x8 kf[4], vf[4];
    {
        const int kl = key < klen ? key : klen - 1;
        const bf16_t* kp = p.qkv + (krow0 + kl) * ld + p.H * HD + h * HD + g * 8;
        const bf16_t* vp = kp + p.H * HD;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = *(const bf16x8*)(kp + kk * 32);
            vf[kk] = *(const bf16x8*)(vp + kk * 32);
        }
    }
    f32x4 dv[8], dk[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // the step table of this sample and the steps' lse pointers, read inside the tile loop, live in LDS (see attn_bwd_dkv_kernel:
    // no compiler-visible vector-memory load may sit between the asm-issued prefetches), and so do the current step's statistics
    __shared__ int s_off[EPI_TMAX], s_n[EPI_TMAX];
    __shared__ const float* s_lsep[EPI_TMAX];
    for (int t = tid; t < T; t += 256) {
        s_off[t] = p.tab[t * B + b]; s_n[t] = p.tab[T * B + t * B + b]; s_lsep[t] = p.lse[t];
    }
    LDS_PTR(float) s_lse = (LDS_PTR(float))(smem + DKV_SLOTS * 2 * TILE);
    LDS_PTR(float) s_ds = s_lse + p.stat_cap;
    __syncthreads();
    auto nvalid = [&](int t) { return s_n[t]; };
    auto norm = [&](int& t, int& q) { while (t < t_end && q * 32 >= nvalid(t)) { ++t; q = 0; } };
    auto stage = [&](int t, int q, int slot) {
        const int n = nvalid(t);
        const long row0 = s_off[t];
        const u32x4 rq = make_desc_u(p.qkv + row0 * ld, (uint32_t)(((long)(n - 1) * ld + 3 * p.H * HD) * 2));
        const u32x4 rdo = make_desc_u(p.dout + row0 * od, (uint32_t)(((long)(n - 1) * od + od) * 2));
        stage_rows<32, 256>(rq, smem + slot * 2 * TILE, q * 32, h * HD, ld, tid);
        stage_rows<32, 256>(rdo, smem + slot * 2 * TILE + TILE, q * 32, h * HD, od, tid);
    };
    int t0 = t_beg, q0 = pre ? 0 : kblk / 32;          // a step's own keys: causal, the first query tile is the one holding key kblk
    norm(t0, q0);
    if (t0 < t_end) {
        int t1 = t0, q1 = q0 + 1;
        norm(t1, q1);
        int t2 = t1, q2 = q1 + 1;
        if (t1 < t_end) norm(t2, q2);
        int t3 = t2, q3 = q2 + 1;
        if (t2 < t_end) norm(t3, q3);
        stage(t0, q0, 0);
        if (t1 < t_end) stage(t1, q1, 1);
        if (t2 < t_end) stage(t2, q2, 2);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { pin(kf[kk]); pin(vf[kk]); }
        wait_stages((t1 < t_end) + (t2 < t_end));
        __builtin_amdgcn_s_barrier();
        int cur = 0, t_stats = -1;
        const int wkey_hi = kblk + wave * 16 + 15;
        while (t0 < t_end) {
            const bool more = t3 < t_end;
            const int n = nvalid(t0);
            const long row0 = s_off[t0];
            if (t0 != t_stats) {
                // a new step: its statistics (n <= stat_cap rows, padded to the tile) into LDS.  Every wave left the previous step's
                // reads behind at the barrier that ended the last tile step; this is the one place the loop drains its prefetches.
                const float* lse2 = s_lsep[t0] + ((long)b * p.H + h) * p.cap + lp;
                const float* dsum = p.dsum + (long)h * p.Rs + (row0 - p.Mp);
                const int npad = (n + 31) & ~31;
                for (int i = tid; i < npad; i += 256) {
                    const int qc = i < n ? i : n - 1;
                    s_lse[i] = lse2[qc];
                    s_ds[i] = dsum[qc];
                }
                __syncthreads();
                t_stats = t0;
            }
            float lq[2][4], dq_[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a4 = *(LDS_PTR(f32x4))(s_lse + q0 * 32 + j * 16 + g * 4), d4 = *(LDS_PTR(f32x4))(s_ds + q0 * 32 + j * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { lq[j][r] = a4[r]; dq_[j][r] = d4[r]; }
            }
            if (more) stage(t3, q3, (cur + 3) & 3);
            LDS_PTR(char) sq = smem + cur * 2 * TILE;
            LDS_PTR(char) sdo = sq + TILE;
            f32x4 s[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // the row-major fragments of two k-steps are requested back to back, then their eight MFMAs run while the later reads are
            // still in flight (second phase: four head-dim tiles at a time).  With two waves per SIMD nothing else hides the LDS
            // latency; hipcc's own schedule keeps ONE read ahead of its MFMA, and sched_barrier stops it from re-interleaving this one.
            // (All sixteen at once, plus the second phase's first half before the softmax arithmetic: 220 VGPRs and no faster.)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                bf16x8 qa[2][2], da[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        qa[kk][j] = frag_rm(sq, j * 16, k2 * 2 + kk, lane);
                        da[kk][j] = frag_rm(sdo, j * 16, k2 * 2 + kk, lane);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[kk][j], kf[k2 * 2 + kk], s[j], 0, 0, 0);
                        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[kk][j], vf[k2 * 2 + kk], dp[j], 0, 0, 0);
                    }
            }
            const bool interior = (q0 * 32 + 31 < n) && (pre ? wkey_hi < lp : q0 * 32 >= wkey_hi);
            f32x4 pv[2], ds[2];
            if (interior) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = fast_exp2(fmaf(s[j][r], p.scale2, -lq[j][r]));
                        pv[j][r] = pe;
                        ds[j][r] = pe * (dp[j][r] - dq_[j][r]);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 * 32 + j * 16 + g * 4 + r;
                        const bool ok = (q < n) && (pre ? key < lp : key <= q);
                        const float pe = ok ? fast_exp2(fmaf(s[j][r], p.scale2, -lq[j][r])) : 0.f;
                        pv[j][r] = pe;
                        ds[j][r] = ok ? pe * (dp[j][r] - dq_[j][r]) : 0.f;
                    }
            }
            const bf16x8 pfrag = pack_frag(pv[0], pv[1]), dsfrag = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int d4 = 0; d4 < 2; ++d4) {
                bf16x8 dot_[4], qt_[4];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    dot_[dd] = frag_tr(sdo, 0, 16, (d4 * 4 + dd) * 16, lane);
                    qt_[dd] = frag_tr(sq, 0, 16, (d4 * 4 + dd) * 16, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    dv[d4 * 4 + dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_[dd], pfrag, dv[d4 * 4 + dd], 0, 0, 0);
                    dk[d4 * 4 + dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_[dd], dsfrag, dk[d4 * 4 + dd], 0, 0, 0);
                }
            }
            wait_stages((t2 < t_end) + (more ? 1 : 0));                       // tile 1 has landed (tiles 2, 3 may be in flight)
            __builtin_amdgcn_s_barrier();
            cur = (cur + 1) & 3;
            t0 = t1; q0 = q1; t1 = t2; q1 = q2; t2 = t3; q2 = q3;
            if (more) { ++q3; norm(t3, q3); }
        }
    }
    if (key >= klen) return;
    if (pre) {
        float* ak = p.kvacc + ((long)b * p.cap + key) * (2L * p.H * HD) + h * HD + g * 4;
        float* av = ak + p.H * HD;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            f32x4 k4 = dk[dt] * p.scale, v4 = dv[dt];
            if (p.kvacc_add) { k4 += *(const f32x4*)(ak + dt * 16); v4 += *(const f32x4*)(av + dt * 16); }
            *(f32x4*)(ak + dt * 16) = k4;
            *(f32x4*)(av + dt * 16) = v4;
        }
        return;
    }
    const int pos = lp + key < p.cap ? lp + key : p.cap - 1;
    bf16_t* kp = p.dqkv + (krow0 + key) * ld + p.H * HD + h * HD + g * 4;
    store_grad_row(kp, dk, p.scale, p.rope_cos ? p.rope_cos + (long)pos * HD + g * 4 : nullptr,
                   p.rope_sin ? p.rope_sin + (long)pos * HD + g * 4 : nullptr);
    store_grad_row(kp + p.H * HD, dv, 1.f, nullptr, nullptr);
}

// grid (B*H, T * nSB): 64 queries of one step per block (wave w: 16 of them); key tiles of 64: the sample's prefix, then the step's own rows
__global__ __launch_bounds__(256) void epi_bwd_dq_kernel(EpiArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LDS_PTR(char) smem = (LDS_PTR(char))smem_raw;
    constexpr int TILE = 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, ld = p.ld, od = p.H * HD;
    const int t = blockIdx.y / p.nSB, q0 = (blockIdx.y % p.nSB) * 64;
    const int n = p.tab[T * p.B + t * p.B + b];
    if (q0 >= n) return;
    const long row0 = p.tab[t * p.B + b];
    const int lp = p.cu[b + 1] - p.cu[b];
    const int qi = lane & 15, g = lane >> 4;
    const int q = q0 + wave * 16 + qi;
    f32x4 dq[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q0 < n) {
        const u32x4 rpre = make_desc_u(p.qkv + (long)p.cu[b] * ld, (uint32_t)(((long)(lp - 1) * ld + 3 * p.H * HD) * 2));
        const u32x4 rstep = make_desc_u(p.qkv + row0 * ld, (uint32_t)(((long)(n - 1) * ld + 3 * p.H * HD) * 2));
        const int kcol = p.H * HD + h * HD, vcol = 2 * p.H * HD + h * HD;
        const int ql = q < n ? q : n - 1;
        bf16x8 qf[4], dof[4];
        {
            const bf16_t* qp = p.qkv + (row0 + ql) * ld + h * HD + g * 8;
            const bf16_t* dp_ = p.dout + (row0 + ql) * od + h * HD + g * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                qf[kk] = *(const bf16x8*)(qp + kk * 32);
                dof[kk] = *(const bf16x8*)(dp_ + kk * 32);
            }
        }
        const float my_lse = p.lse[t][((long)b * p.H + h) * p.cap + lp + ql];
        const float my_ds = p.dsum[(long)h * p.Rs + (row0 - p.Mp) + ql];
        const int q_hi = q0 + 63 < n - 1 ? q0 + 63 : n - 1;
        const int npt = (lp + 63) / 64, total = npt + q_hi / 64 + 1;
        auto stage = [&](int i, int buf) {
            if (i < npt) {
                stage_rows<64, 256>(rpre, smem + buf * 2 * TILE, i * 64, kcol, ld, tid);
                stage_rows<64, 256>(rpre, smem + buf * 2 * TILE + TILE, i * 64, vcol, ld, tid);
            } else {
                stage_rows<64, 256>(rstep, smem + buf * 2 * TILE, (i - npt) * 64, kcol, ld, tid);
                stage_rows<64, 256>(rstep, smem + buf * 2 * TILE + TILE, (i - npt) * 64, vcol, ld, tid);
            }
        };
        stage(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { pin(qf[kk]); pin(dof[kk]); }
        float lse_q = my_lse, ds_q = my_ds;
        pin(lse_q); pin(ds_q);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        const int wq_lo = q0 + wave * 16, wq_hi = wq_lo + 15;
        for (int i = 0; i < total; ++i) {
            if (i + 1 < total) stage(i + 1, cur ^ 1);
            LDS_PTR(char) sk = smem + cur * 2 * TILE;
            LDS_PTR(char) sv = sk + TILE;
            const bool is_pre = i < npt;
            const int kt = is_pre ? i : i - npt;
            f32x4 s[4], dp[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { s[a] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 ka[4], va[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    ka[a] = frag_rm(sk, a * 16, kk, lane);
                    va[a] = frag_rm(sv, a * 16, kk, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    s[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[a], qf[kk], s[a], 0, 0, 0);
                    dp[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[a], dof[kk], dp[a], 0, 0, 0);
                }
            }
            const bool interior = (wq_hi < n) && (is_pre ? kt * 64 + 63 < lp : kt * 64 + 63 <= wq_lo);
            if (interior) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = fast_exp2(fmaf(s[a][r], p.scale2, -lse_q));
                        s[a][r] = pe * (dp[a][r] - ds_q);
                    }
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + a * 16 + g * 4 + r;
                        const bool ok = (q < n) && (is_pre ? key < lp : key <= q);
                        const float pe = ok ? fast_exp2(fmaf(s[a][r], p.scale2, -lse_q)) : 0.f;
                        s[a][r] = ok ? pe * (dp[a][r] - ds_q) : 0.f;
                    }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16x8 dsf = pack_frag(s[2 * a], s[2 * a + 1]);
                bf16x8 kt_[8];
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) kt_[dt] = frag_tr(sk, a * 32, a * 32 + 16, dt * 16, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt], dsf, dq[dt], 0, 0, 0);
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
    }
    if (q >= n) return;
    const int pos = lp + q < p.cap ? lp + q : p.cap - 1;
    bf16_t* qp = p.dqkv + (row0 + q) * ld + h * HD + g * 4;
    store_grad_row(qp, dq, p.scale, p.rope_cos ? p.rope_cos + (long)pos * HD + g * 4 : nullptr,
                   p.rope_sin ? p.rope_sin + (long)pos * HD + g * 4 : nullptr);
}

int set_lds(const void* fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? NV_OK : NV_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int nv_attn_fwd_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S, int H, int head_dim,
                     int q_row_min, void* stream) {
    if (!qkv || !out || !lse2 || !kv_start) return NV_ERR_ARG;
    if (head_dim != HD || (q_row_min & 127) || q_row_min < 0 || (S > 0 && q_row_min >= S)) return NV_ERR_SHAPE;
    if (B == 0 || S == 0) return NV_OK;
    static bool once = false;
    if (!once) { if (set_lds((const void*)attn_fwd_kernel<false>, 65536)) return NV_ERR_LAUNCH; once = true; }
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse2 = lse2; p.kv_start = kv_start;
    p.B = B; p.S = S; p.Sst = S; p.H = H; p.ld = 3 * H * HD; p.q_row_min = q_row_min;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    NV_LAUNCH(attn_fwd_kernel<false>, dim3(B * H, (S - q_row_min + 127) / 128), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// The same forward over a KV-cache layout: sample b's rows start at b*S_stride (S_stride >= S = the longest valid
// length); out and lse2 use the same sample stride.  Inference only (prefix reuse / generation).
int nv_attn_fwd_strided_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S, int S_stride, int H,
                             int head_dim, int q_row_min, void* stream) {
    if (!qkv || !out || !lse2 || !kv_start) return NV_ERR_ARG;
    if (head_dim != HD || (q_row_min & 127) || q_row_min < 0 || (S > 0 && q_row_min >= S) || S_stride < S) return NV_ERR_SHAPE;
    if (B == 0 || S == 0) return NV_OK;
    static bool once = false;
    if (!once) { if (set_lds((const void*)attn_fwd_kernel<false>, 65536)) return NV_ERR_LAUNCH; once = true; }
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse2 = lse2; p.kv_start = kv_start;
    p.B = B; p.S = S; p.Sst = S_stride; p.H = H; p.ld = 3 * H * HD; p.q_row_min = q_row_min;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    NV_LAUNCH(attn_fwd_kernel<false>, dim3(B * H, (S - q_row_min + 127) / 128), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// Forward attention of ALL steps of a prefix-reuse episode in one launch, reading the episode row buffers in place (see
// epi_fwd_kernel): qkv [rows, 3*H*128] post-RoPE (prefix rows of sample b at [cu[b], cu[b+1]), step t's rows of sample b at
// [tab[t*B+b], + tab[T*B + t*B + b])), out [rows, H*128] written at the steps' rows, lse_ptrs[t] -> fp32 [B, H, cap] of step t written
// at cache position prefix_len + j.  n_max = the longest per-sample row count of a step.  Bit-identical to scattering each step
// into the K/V-cache layout + nv_attn_fwd_strided_bf16 + gathering back.
int nv_attn_fwd_episode_bf16(const void* qkv, void* out, const void* lse_ptrs, const int* cu, const int* tab, int T, int B, int H,
                             int head_dim, int cap, int n_max, long rows, void* stream) {
    if (!qkv || !out || !lse_ptrs || !cu || !tab || T < 0 || B <= 0 || H <= 0 || cap <= 0 || rows <= 0) return NV_ERR_ARG;
    if (head_dim != HD) return NV_ERR_SHAPE;
    if (T == 0 || n_max <= 0) return NV_OK;
    const long bytes = rows * 3L * H * HD * 2;
    if (bytes > 0xffffffffL) return NV_ERR_SHAPE;
    if ((long)T * ((n_max + 127) / 128) > 65535) return NV_ERR_SHAPE;      // grid.y limit: the caller takes the per-step path
    static bool once = false;
    if (!once) { if (set_lds((const void*)epi_fwd_kernel, 65536)) return NV_ERR_LAUNCH; once = true; }
    EpiFwdArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse = (float* const*)lse_ptrs; p.cu = cu; p.tab = tab;
    p.T = T; p.B = B; p.H = H; p.ld = 3 * H * HD; p.cap = cap; p.QB = (n_max + 127) / 128;
    p.span = (uint32_t)bytes;
    p.scale2 = (1.f / sqrtf((float)HD)) * 1.4426950408889634f;
    NV_LAUNCH(epi_fwd_kernel, dim3(B * H, T * p.QB), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// Decode attention: query r = the q slice of cache row crow[r] of SAMPLE r (one new token per sample), keys/values = cache rows
// r*cap + [0, pos[r]], out [M, H*128] compact.  pos / crow are device arrays: nothing in the launch depends on the lengths.
int nv_attn_decode_bf16(const void* kv, const int* crow, const int* pos, void* out, int M, int H, int head_dim, int cap, void* stream) {
    if (!kv || !crow || !pos || !out || cap <= 0) return NV_ERR_ARG;
    if (head_dim != HD) return NV_ERR_SHAPE;
    if (M == 0) return NV_OK;
    const float scale2 = (1.f / sqrtf((float)HD)) * 1.4426950408889634f;
    NV_LAUNCH(attn_decode_kernel, dim3(H, M), dim3(DEC_WAVES * 64), 0, (hipStream_t)stream, (const bf16_t*)kv, crow, pos, (bf16_t*)out, H, cap,
              scale2);
    return nv_check_launch();
}

// The same over a K/V cache whose current length and first computed query row live in DEVICE memory: dyn = {S, q_row_min}
// (q_row_min a multiple of 128, < S <= S_stride).  The grid covers the whole capacity; query blocks at or beyond S exit at once.
int nv_attn_fwd_strided_dyn_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S_stride, int H, int head_dim,
                                 const int* dyn, void* stream) {
    if (!qkv || !out || !lse2 || !kv_start || !dyn) return NV_ERR_ARG;
    if (head_dim != HD || S_stride <= 0) return NV_ERR_SHAPE;
    if (B == 0) return NV_OK;
    static bool once = false;
    if (!once) { if (set_lds((const void*)attn_fwd_kernel<false>, 65536)) return NV_ERR_LAUNCH; once = true; }
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse2 = lse2; p.kv_start = kv_start;
    p.B = B; p.S = S_stride; p.Sst = S_stride; p.H = H; p.ld = 3 * H * HD; p.q_row_min = 0; p.dyn = dyn;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    NV_LAUNCH(attn_fwd_kernel<false>, dim3(B * H, (S_stride + 127) / 128), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// Packed ("varlen") rows: sample b = rows [cu[b], cu[b+1]) of qkv / out (cu: int32 [B+1] on the device), S_max = the
// longest sample (lse2 is [B,H,S_max]); no padding keys exist, pos0[b] = position of the sample's first token (only the
// backward's fused RoPE^T reads it).  q_row_min >= 0 as above, or -1: each sample's own last 128-row block (pruned last
// decoder layer).  Removes the left-padding rows of a batch from every row kernel and GEMM of the LM.
int nv_attn_fwd_varlen_bf16(const void* qkv, void* out, float* lse2, const int* cu, const int* pos0, int B, int S_max, int H,
                            int head_dim, int q_row_min, void* stream) {
    if (!qkv || !out || !lse2 || !cu || !pos0) return NV_ERR_ARG;
    if (head_dim != HD || (q_row_min >= 0 && (q_row_min & 127)) || q_row_min < -1 || (S_max > 0 && q_row_min >= S_max)) return NV_ERR_SHAPE;
    if (B == 0 || S_max == 0) return NV_OK;
    static bool once = false;
    if (!once) { if (set_lds((const void*)attn_fwd_kernel<false>, 65536)) return NV_ERR_LAUNCH; once = true; }
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse2 = lse2; p.kv_start = pos0; p.cu = cu;
    p.B = B; p.S = S_max; p.Sst = S_max; p.H = H; p.ld = 3 * H * HD; p.q_row_min = q_row_min;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    const int qblocks = q_row_min < 0 ? 1 : (S_max - q_row_min + 127) / 128;
    NV_LAUNCH(attn_fwd_kernel<false>, dim3(B * H, qblocks), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// PARITY INSTRUMENT (tests only; never called by navillm_amd's forward): the same attention with HF's eager-attention rounding
// points (attn_fwd_kernel<true>, two passes).  cu == NULL: padded [B, S] layout with kv_start; else packed rows (kv_start =
// pos0).  Same outputs / layouts as nv_attn_fwd_bf16 resp. nv_attn_fwd_varlen_bf16.
int nv_attn_fwd_hfround_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, const int* cu, int B, int S, int H,
                             int head_dim, int q_row_min, void* stream) {
    if (!qkv || !out || !lse2 || !kv_start) return NV_ERR_ARG;
    if (head_dim != HD || (q_row_min >= 0 && (q_row_min & 127)) || q_row_min < (cu ? -1 : 0) || (S > 0 && q_row_min >= S)) return NV_ERR_SHAPE;
    if (B == 0 || S == 0) return NV_OK;
    static bool once = false;
    if (!once) { if (set_lds((const void*)attn_fwd_kernel<true>, 65536)) return NV_ERR_LAUNCH; once = true; }
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse2 = lse2; p.kv_start = kv_start; p.cu = cu;
    p.B = B; p.S = S; p.Sst = S; p.H = H; p.ld = 3 * H * HD; p.q_row_min = q_row_min;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    const int qblocks = q_row_min < 0 ? 1 : (S - q_row_min + 127) / 128;
    NV_LAUNCH(attn_fwd_kernel<true>, dim3(B * H, qblocks), dim3(256), 65536, (hipStream_t)stream, p);
    return nv_check_launch();
}

// dsum: workspace [B,H,S] fp32 (nv_attn_bwd_workspace_bytes)
size_t nv_attn_bwd_workspace_bytes(int B, int S, int H) { return (size_t)B * S * H * sizeof(float); }

static int attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                         void* workspace, int B, int S, int H, int head_dim, int q_row_min, const void* rope_cos,
                         const void* rope_sin, void* stream, const int* cu = nullptr, long rows = -1, int Sst = 0, float* kvacc = nullptr,
                         const int* kvacc_len = nullptr, int kvacc_first = 0) {
    if (!qkv || !out || !dout || !lse2 || !kv_start || !dqkv || !workspace) return NV_ERR_ARG;
    if (head_dim != HD || (q_row_min >= 0 && (q_row_min & 127)) || q_row_min < (cu ? -1 : 0) || (S > 0 && q_row_min >= S)) return NV_ERR_SHAPE;
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return NV_ERR_ARG;
    if (B == 0 || S == 0) return NV_OK;
    // measurement/test knob: 1 (default) = 16 rows per wave, 2 = 32 rows per wave.  Measured at B=8, S=656, H=32: the
    // 32-row form halves the LDS fragment reads per MFMA but runs 539 vs 344 us -- with one barrier-synchronised tile
    // step per iteration these kernels live on occupancy, which the wider form halves.
    const char* ev = getenv("NV_ATTN_BWD_VARIANT");
    const int variant = ev ? atoi(ev) : 1;
    static bool once = false;
    if (!once) {
        if (set_lds((const void*)attn_bwd_dkv_kernel<1>, 77824) || set_lds((const void*)attn_bwd_dq_kernel<1>, 65536) ||
            set_lds((const void*)attn_bwd_dkv_kernel<2>, 77824) || set_lds((const void*)attn_bwd_dq_kernel<2>, 65536))
            return NV_ERR_LAUNCH;
        once = true;
    }
    hipStream_t st = (hipStream_t)stream;
    float* dsum = (float*)workspace;
    if (Sst <= 0) Sst = S;
    // padded / strided (K/V-cache) layout: dsum is indexed like lse2, [B, H, Sst]; only the query rows [q_row_min, S) of a sample are
    // read by the kernels below
    const int q_lo = cu ? 0 : q_row_min, nq = cu ? 0 : S - q_row_min;
    if (rows < 0) rows = (long)B * nq;
    const long items = rows * H;
    NV_LAUNCH(attn_bwd_prep_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, st, (const bf16_t*)dout,
                       (const bf16_t*)out, dsum, cu, rows, B, cu ? S : Sst, H, q_lo, nq);
    AttnArgs p{};
    p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.lse2 = (float*)lse2; p.dsum = dsum; p.dqkv = (bf16_t*)dqkv;
    p.kv_start = kv_start; p.cu = cu; p.B = B; p.S = S; p.Sst = Sst; p.H = H; p.ld = 3 * H * HD; p.q_row_min = q_row_min;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    p.rope_cos = (const bf16_t*)rope_cos; p.rope_sin = (const bf16_t*)rope_sin;
    p.kvacc = kvacc; p.kvacc_len = kvacc_len; p.kvacc_first = kvacc_first;
    const int scap = (S + 31) & ~31;
    p.stat_cap = 8 * scap <= 12288 ? scap : 0;                // dK/dV: lse + dsum of one (sample, head) in LDS when they fit (S <= 1536)
    const int dkv_lds = 65536 + 8 * p.stat_cap;               // 4 slots x (Q, dO) x 8 KiB, then the statistics: two blocks per CU
    // with q_row_min > 0 only those query rows carry gradient: dK/dV still cover every key, dQ rows below
    // q_row_min are NOT written (the caller zero-fills them)
    if (variant == 1) {
        NV_LAUNCH(attn_bwd_dkv_kernel<1>, dim3(B * H, (S + 63) / 64), dim3(256), dkv_lds, st, p);
        NV_LAUNCH(attn_bwd_dq_kernel<1>, dim3(B * H, q_row_min < 0 ? 2 : (S - q_row_min + 63) / 64), dim3(256), 65536, st, p);
    } else {
        NV_LAUNCH(attn_bwd_dkv_kernel<2>, dim3(B * H, (S + 127) / 128), dim3(256), dkv_lds, st, p);
        NV_LAUNCH(attn_bwd_dq_kernel<2>, dim3(B * H, q_row_min < 0 ? 1 : (S - q_row_min + 127) / 128), dim3(256), 65536, st, p);
    }
    return nv_check_launch();
}

int nv_attn_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                     void* workspace, int B, int S, int H, int head_dim, int q_row_min, void* stream) {
    return attn_bwd_impl(qkv, out, dout, lse2, kv_start, dqkv, workspace, B, S, H, head_dim, q_row_min, nullptr, nullptr, stream);
}

// The same with the transpose of RoPE applied to dQ and dK as they are written (cos/sin: the [maxS, 128] bf16 tables of
// nv_rope_bf16; position of a row = its index within the sample): saves the separate in-place pass over dqkv.
int nv_attn_bwd_rope_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                          void* workspace, const void* rope_cos, const void* rope_sin, int B, int S, int H, int head_dim,
                          int q_row_min, void* stream) {
    if (!rope_cos || !rope_sin) return NV_ERR_ARG;
    return attn_bwd_impl(qkv, out, dout, lse2, kv_start, dqkv, workspace, B, S, H, head_dim, q_row_min, rope_cos, rope_sin, stream);
}

// backward over the K/V-cache layout of nv_attn_fwd_strided_bf16 (sample b at rows b*S_stride .., valid length <= S; out, dout and
// dqkv use the same sample stride; workspace = nv_attn_bwd_workspace_bytes(B, S_stride, H)).  Queries >= q_row_min carry gradient
// (dO of every other row must be zero); dK/dV are written for every key row < S of every sample, dQ for rows >= q_row_min; no
// RoPE^T (the gradients stay in the rotated frame of the cache).  Training with a cached prompt prefix (navillm_amd/episode.py).
int nv_attn_bwd_strided_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                             void* workspace, int B, int S, int S_stride, int H, int head_dim, int q_row_min, void* stream) {
    if (S_stride < S) return NV_ERR_SHAPE;
    return attn_bwd_impl(qkv, out, dout, lse2, kv_start, dqkv, workspace, B, S, H, head_dim, q_row_min, nullptr, nullptr, stream, nullptr,
                         -1, S_stride);
}

// The same with the gradients of each sample's first prefix_len[b] key rows (a cached prompt prefix) accumulated in fp32 into
// kv_acc [B*S_stride, 2*H*head_dim] (`first` != 0: stored) instead of written to dqkv as bf16: replaces nv_kv_grad_accum_f32 /
// nv_kv_grad_set_f32 over those rows.  dqkv's K/V columns of the prefix rows are left untouched.
int nv_attn_bwd_strided_kvacc_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                                   void* workspace, float* kv_acc, const int* prefix_len, int first, int B, int S, int S_stride, int H,
                                   int head_dim, int q_row_min, void* stream) {
    if (S_stride < S || !kv_acc || !prefix_len) return NV_ERR_SHAPE;
    return attn_bwd_impl(qkv, out, dout, lse2, kv_start, dqkv, workspace, B, S, H, head_dim, q_row_min, nullptr, nullptr, stream, nullptr,
                         -1, S_stride, kv_acc, prefix_len, first);
}

// backward over packed rows (see nv_attn_fwd_varlen_bf16); rope_cos/rope_sin optional (both or neither); `rows` = cu[B]
int nv_attn_bwd_varlen_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* cu, const int* pos0,
                            void* dqkv, void* workspace, const void* rope_cos, const void* rope_sin, int B, int S_max, long rows,
                            int H, int head_dim, int q_row_min, void* stream) {
    if (!cu || rows < 0) return NV_ERR_ARG;
    return attn_bwd_impl(qkv, out, dout, lse2, pos0, dqkv, workspace, B, S_max, H, head_dim, q_row_min, rope_cos, rope_sin, stream, cu,
                         rows);
}

// Attention backward of ALL the steps of a prefix-reuse episode for one layer, in place on the episode's row buffers (see EpiArgs):
// qkv / dqkv [R, 3*H*head_dim], out / dout [R, H*head_dim]; rows [0, Mp) are the packed prompt prefixes (cu [B+1]), then the T step
// blocks, packed too, described by tab (device int32: off[T*B] = first row of step t / sample b | n[T*B] = its rows; together they
// tile [Mp, R)); lse_ptrs = T device pointers to the steps' lse2 [B, H, cap].  Writes dqkv rows [Mp, R) (dQ and the steps' own dK|dV,
// through RoPE^T when the tables are given: position = prefix_len + j) and STORES the fp32 sum over the steps of the prefix rows'
// dK|dV in kv_acc [B*cap, 2*H*head_dim] (row b*cap + key).  workspace: (R - Mp) * H floats.  Lp_max / N_max: the longest prefix /
// the most rows a sample has in one step.
// `accumulate` != 0: the prefix rows' dK|dV are ADDED to kv_acc instead of stored -- segments of a long episode (navillm_amd/episode.py
// flushes the deferred backward of the steps so far when their rows no longer fit; the first segment stores, later ones add).
int nv_attn_bwd_episode_acc_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, void* workspace, const void* lse_ptrs,
                                 const int* cu, const int* tab, float* kv_acc, const void* rope_cos, const void* rope_sin, int T, int B, int H,
                                 int head_dim, int cap, int Mp, long R, int Lp_max, int N_max, int accumulate, void* stream);
int nv_attn_bwd_episode_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, void* workspace, const void* lse_ptrs,
                             const int* cu, const int* tab, float* kv_acc, const void* rope_cos, const void* rope_sin, int T, int B, int H,
                             int head_dim, int cap, int Mp, long R, int Lp_max, int N_max, void* stream) {
    return nv_attn_bwd_episode_acc_bf16(qkv, out, dout, dqkv, workspace, lse_ptrs, cu, tab, kv_acc, rope_cos, rope_sin, T, B, H, head_dim, cap, Mp,
                                        R, Lp_max, N_max, 0, stream);
}
int nv_attn_bwd_episode_acc_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, void* workspace, const void* lse_ptrs,
                                 const int* cu, const int* tab, float* kv_acc, const void* rope_cos, const void* rope_sin, int T, int B, int H,
                                 int head_dim, int cap, int Mp, long R, int Lp_max, int N_max, int accumulate, void* stream) {
    if (!qkv || !out || !dout || !dqkv || !workspace || !lse_ptrs || !cu || !tab || !kv_acc) return NV_ERR_ARG;
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return NV_ERR_ARG;
    if (head_dim != HD || T < 0 || B <= 0 || H <= 0 || cap <= 0 || Mp < 0 || R < Mp || Lp_max <= 0 || Lp_max > cap || N_max < 0) return NV_ERR_SHAPE;
    if (T == 0 || R == Mp || N_max == 0) return NV_OK;
    const int scap = (N_max + 31) & ~31;
    if (T > EPI_TMAX || 8 * scap > 12288) return NV_ERR_SHAPE;    // step table / statistics rows beyond the LDS budget (N_max <= 1536)
    static bool once = false;
    if (!once) {
        if (set_lds((const void*)epi_bwd_dkv_kernel, 77824) || set_lds((const void*)epi_bwd_dq_kernel, 65536)) return NV_ERR_LAUNCH;
        once = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const int od = H * HD;
    const long Rs = R - Mp;
    float* dsum = (float*)workspace;
    NV_LAUNCH(attn_bwd_prep_kernel, dim3((unsigned)((Rs * H + 15) / 16)), dim3(256), 0, st, (const bf16_t*)dout + (long)Mp * od,
              (const bf16_t*)out + (long)Mp * od, dsum, (const int*)nullptr, Rs, 1, (int)Rs, H, 0, (int)Rs);
    EpiArgs p{};
    p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.dqkv = (bf16_t*)dqkv; p.dsum = dsum;
    p.lse = (const float* const*)lse_ptrs; p.cu = cu; p.tab = tab; p.kvacc = kv_acc;
    p.rope_cos = (const bf16_t*)rope_cos; p.rope_sin = (const bf16_t*)rope_sin;
    p.T = T; p.B = B; p.H = H; p.ld = 3 * H * HD; p.cap = cap; p.Mp = Mp; p.Rs = (int)Rs;
    p.nPB = (Lp_max + 63) / 64; p.nSB = (N_max + 63) / 64; p.stat_cap = scap; p.kvacc_add = accumulate ? 1 : 0;
    p.scale = 1.f / sqrtf((float)HD); p.scale2 = p.scale * 1.4426950408889634f;
    NV_LAUNCH(epi_bwd_dkv_kernel, dim3(B * H, p.nPB + T * p.nSB), dim3(256), 65536 + 8 * scap, st, p);
    NV_LAUNCH(epi_bwd_dq_kernel, dim3(B * H, T * p.nSB), dim3(256), 65536, st, p);
    return nv_check_launch();
}


}  // extern "C"
