// nv_decoder_*: the inference-time decoder stack as ONE native call ("runtime in native code").
//
// A K/V-reuse navigation step or a decode step of generation pushes a few new token rows through all L decoder layers:
// 9 kernel launches per layer.  Driven from Python that is ~290 ctypes calls + as many tensor allocations per step, and the
// host -- not the GPU -- set the pace (262 nav-steps/s in round 1 for ~800 new rows).  Here the layer loop lives in C++:
// one call walks a table of layer pointers and enqueues every launch of the step on the given stream, with all intermediates
// carved from one caller-provided workspace.  The kernels are the same C-ABI entry points the training path uses
// (nv_rmsnorm_fwd_bf16, nv_gemm_bf16_ws / nv_gemv_bf16, nv_rope_rows_bf16, nv_scatter/gather_rows_bf16,
// nv_attn_fwd_strided_bf16, nv_swiglu_fwd_bf16) plus the weight-only fp8 pair (nv_fp8_dequant_rows + GEMM, nv_gemv_fp8w).
// Reference counterpart: HF LlamaModel.forward with past_key_values, reached from models/modified_lm.py:184-199 (generation)
// -- the per-step prompt recompute of tasks/agents/mp3d_agent.py:726 has no cache at all.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/navillm_hip.h"

namespace {

struct Layer {
    const void *norm1 = nullptr, *norm2 = nullptr;
    const void* w[4] = {nullptr, nullptr, nullptr, nullptr};        // bf16 operands: qkv [3d,d], o [d,d], gate|up [2ff,d], down [d,ff]
    const void* q[4] = {nullptr, nullptr, nullptr, nullptr};        // or e4m3fn codes ...
    const float* s[4] = {nullptr, nullptr, nullptr, nullptr};       // ... + per-output-channel scales
    void* kv = nullptr;                                             // this layer's packed post-RoPE qkv cache [B*cap + 1, 3d]
};

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

struct nv_decoder {
    int L, d, H, hd, ff;
    float eps;
    std::vector<Layer> layers;
    const void *rope_cos = nullptr, *rope_sin = nullptr, *final_norm = nullptr;
    void* gemm_ws = nullptr;        // zero-filled split-K workspace of nv_gemm_bf16_ws (may be NULL)
    void* fp8_scratch = nullptr;    // bf16 panel for de-quantised operands (needed when a layer carries codes)
    int fp8_gemm_mode = 0;          // nv_gemm_fp8w mode of THIS decoder's few-hundred-row GEMMs (7 | 9; 0 = the process default)
    // weight-only fp8, memory-lean AND overlapped (round 3): two panels; the NEXT Linear's operand is de-quantised on a side stream
    // while the current GEMM runs (the GEMM is MFMA-bound and leaves most of the HBM bandwidth idle; the pre-pass is pure bandwidth)
    void* panel[2] = {nullptr, nullptr};
    hipStream_t side = nullptr;
    hipEvent_t ev_entry = nullptr, ev_dq[2] = {nullptr, nullptr}, ev_use[2] = {nullptr, nullptr};
};

extern "C" {

nv_decoder* nv_decoder_create(int L, int d, int H, int head_dim, int ff, float eps) {
    if (L <= 0 || d <= 0 || H <= 0 || head_dim != 128 || H * head_dim != d || ff <= 0) return nullptr;
    nv_decoder* p = new (std::nothrow) nv_decoder();
    if (!p) return nullptr;
    p->L = L; p->d = d; p->H = H; p->hd = head_dim; p->ff = ff; p->eps = eps;
    p->layers.resize(L);
    return p;
}

void nv_decoder_destroy(nv_decoder* p) {
    if (!p) return;
    if (p->ev_entry) (void)hipEventDestroy(p->ev_entry);
    for (int k = 0; k < 2; ++k) {
        if (p->ev_dq[k]) (void)hipEventDestroy(p->ev_dq[k]);
        if (p->ev_use[k]) (void)hipEventDestroy(p->ev_use[k]);
    }
    delete p;
}

// weight-only fp8 without resident bf16 operands: give the decoder two bf16 panels (each as large as the largest operand) and a
// side stream, and nv_decoder_extend de-quantises the next Linear's operand there while the current GEMM runs on the caller's
// stream (steps of more than 16 rows; the few-row steps read the codes directly).  Everything the call enqueues is ordered
// after the caller's earlier work on `stream` and completes in `stream` order, as without the side stream.
int nv_decoder_set_fp8_overlap(nv_decoder* p, void* panel_a, void* panel_b, void* side_stream) {
    if (!p || !panel_a || !panel_b || panel_a == panel_b || !side_stream) return NV_ERR_ARG;
    if (!p->ev_entry) {
        hipEvent_t* evs[5] = {&p->ev_entry, &p->ev_dq[0], &p->ev_dq[1], &p->ev_use[0], &p->ev_use[1]};
        for (hipEvent_t* e : evs)
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return NV_ERR_LAUNCH;
    }
    p->panel[0] = panel_a; p->panel[1] = panel_b; p->side = (hipStream_t)side_stream;
    return NV_OK;
}

// how this decoder's tile GEMMs on fp8 codes form their operands: 7 = bf16(s*q) (bit-identical to the pre-pass), 9 = codes converted
// unscaled + s[n] on the fp32 accumulator, 0 = the process default (nv_gemm_fp8w_default_mode).  Per object: two models with
// different modes in one process keep their own.
int nv_decoder_set_fp8_gemm_mode(nv_decoder* p, int mode) {
    if (!p || !(mode == 0 || mode == 7 || mode == 9)) return NV_ERR_ARG;
    p->fp8_gemm_mode = mode;
    return NV_OK;
}

// weights of layer i.  kind: 0 qkv, 1 o, 2 gate|up, 3 down.  Either `w` (bf16 [N,K]) or `codes` + `scales` (weight-only fp8).
int nv_decoder_set_weight(nv_decoder* p, int i, int kind, const void* w, const void* codes, const float* scales) {
    if (!p || i < 0 || i >= p->L || kind < 0 || kind > 3) return NV_ERR_ARG;
    if (!w && !(codes && scales)) return NV_ERR_ARG;
    p->layers[i].w[kind] = w; p->layers[i].q[kind] = codes; p->layers[i].s[kind] = scales;
    return NV_OK;
}

int nv_decoder_set_layer(nv_decoder* p, int i, const void* input_norm, const void* post_attn_norm, void* kv_cache) {
    if (!p || i < 0 || i >= p->L || !input_norm || !post_attn_norm || !kv_cache) return NV_ERR_ARG;
    p->layers[i].norm1 = input_norm; p->layers[i].norm2 = post_attn_norm; p->layers[i].kv = kv_cache;
    return NV_OK;
}

int nv_decoder_set_shared(nv_decoder* p, const void* rope_cos, const void* rope_sin, const void* final_norm, void* gemm_workspace,
                          void* fp8_scratch) {
    if (!p || !rope_cos || !rope_sin || !final_norm) return NV_ERR_ARG;
    p->rope_cos = rope_cos; p->rope_sin = rope_sin; p->final_norm = final_norm; p->gemm_ws = gemm_workspace; p->fp8_scratch = fp8_scratch;
    return NV_OK;
}

// bytes of workspace for steps of up to max_rows new token rows
size_t nv_decoder_workspace_bytes(const nv_decoder* p, int max_rows) {
    if (!p || max_rows <= 0) return 0;
    const size_t M = (size_t)max_rows, d = p->d, ff = p->ff;
    // n | attn | x1 | out0 | out1 : [M,d] each;  qkv [M,3d];  gu [M,2ff];  h [M,ff];  rstd [M] fp32
    return align_up(M * d * 2) * 5 + align_up(M * 3 * d * 2) + align_up(M * 2 * ff * 2) + align_up(M * ff * 2) + align_up(M * 4) + 256;
}

static int linear(const nv_decoder* p, const Layer& ly, int kind, const void* x, void* out, const void* R, int M, int N, int K,
                  void* stream) {
    const int epi = R ? 2 : 0;                                       // residual add or plain store
    if (ly.q[kind]) {
        if (M <= 16 && (K & 63) == 0) return nv_gemv_fp8w(x, ly.q[kind], ly.s[kind], out, R, M, N, K, K, K, N, N, epi, stream);
        if (ly.w[kind])                                                // the de-quantised operand is resident: no pre-pass
            return nv_gemm_bf16_ws(0, x, ly.w[kind], out, R, M, N, K, K, K, N, N, epi, 0, p->gemm_ws, stream);
        // round 4: few-hundred-row steps multiply with the codes themselves (weight tile DMA'd as bytes, converted on the MFMA
        // fragment path); NV_ERR_SHAPE = not a cut-off-tile shape -> the de-quantisation pre-pass + the bf16 GEMM
        static const bool tile_fp8 = [] { const char* e = getenv("NAVILLM_FP8_TILE_GEMM"); return !e || e[0] != '0'; }();
        if (tile_fp8) {
            const int rc8 = nv_gemm_fp8w(x, ly.q[kind], ly.s[kind], out, R, M, N, K, K, K, N, N, epi, p->fp8_gemm_mode, 0, p->gemm_ws, stream);
            if (rc8 != NV_ERR_SHAPE) return rc8;
        }
        if (!p->fp8_scratch) return NV_ERR_ARG;
        int rc = nv_fp8_dequant_rows(ly.q[kind], ly.s[kind], p->fp8_scratch, N, K, K, K, stream);
        if (rc) return rc;
        return nv_gemm_bf16_ws(0, x, p->fp8_scratch, out, R, M, N, K, K, K, N, N, epi, 0, p->gemm_ws, stream);
    }
    if (M <= 16 && (K & 31) == 0) return nv_gemv_bf16(x, ly.w[kind], out, R, M, N, K, K, K, N, N, epi, stream);
    return nv_gemm_bf16_ws(0, x, ly.w[kind], out, R, M, N, K, K, K, N, N, epi, 0, p->gemm_ws, stream);
}

namespace {

// the Linears of a step in execution order: k = 4 * layer + kind
struct Fp8Pipe {
    const nv_decoder* p;
    hipStream_t main;
    bool on;
    int dims[4][2];          // N, K of each kind

    int prefetch(int k) const {          // side stream: operand k -> panel k & 1, once the GEMM that last read that panel is done
        if (k >= 4 * p->L) return NV_OK;
        const Layer& ly = p->layers[k >> 2];
        const int kind = k & 3, pn = k & 1;
        if (!ly.q[kind] || ly.w[kind]) return NV_OK;
        if (k >= 2 && hipStreamWaitEvent(p->side, p->ev_use[pn], 0) != hipSuccess) return NV_ERR_LAUNCH;
        const int N = dims[kind][0], K = dims[kind][1];
        int rc = nv_fp8_dequant_rows(ly.q[kind], ly.s[kind], p->panel[pn], N, K, K, K, (void*)p->side);
        if (rc) return rc;
        return hipEventRecord(p->ev_dq[pn], p->side) == hipSuccess ? NV_OK : NV_ERR_LAUNCH;
    }
    // main stream: GEMM k on its panel (de-quantised by prefetch(k)), then the operand after next starts
    int linear(int k, const void* x, void* out, const void* R, int M) const {
        const Layer& ly = p->layers[k >> 2];
        const int kind = k & 3, pn = k & 1, N = dims[kind][0], K = dims[kind][1];
        const int epi = R ? 2 : 0;
        if (!ly.q[kind] || ly.w[kind]) return NV_ERR_ARG;
        if (hipStreamWaitEvent(main, p->ev_dq[pn], 0) != hipSuccess) return NV_ERR_LAUNCH;
        int rc = nv_gemm_bf16_ws(0, x, p->panel[pn], out, R, M, N, K, K, K, N, N, epi, 0, p->gemm_ws, (void*)main);
        if (rc) return rc;
        if (hipEventRecord(p->ev_use[pn], main) != hipSuccess) return NV_ERR_LAUNCH;
        return prefetch(k + 2);
    }
};

}  // namespace

// One incremental step over the K/V cache (navillm_amd/kvcache.py::KVCacheLM.extend is the host side):
//   x        [M, d] bf16   embeddings of the new rows, any order (the host packs them sample after sample); M == B is taken to mean
//                          ONE new row per sample, row r = sample r (the decode step)
//   pos      [M] int32     position of every row;  crow [M]: cache row it is scattered to (a junk row for padding);
//   grow     [M] int32     cache row whose attention output it reads back
//   kv0      [B] int32     zeros (samples are left-aligned in the cache);  attn_buf [B*cap, d] bf16, lse [B, H, cap] fp32
//   last     [B] int32     block row of each sample's last token -> hs_out [B, d] = final-norm hidden state of those rows
//   hs_all   optional [M, d]: final-norm hidden states of ALL block rows (generation reads none; tests may)
//   dyn      optional DEVICE {Lmax, q_row_min}: overrides the two host values (a step replayed from a hipGraph)
int nv_decoder_extend(const nv_decoder* p, const void* x_in, const int* pos, const int* crow, const int* grow, const int* kv0,
                      void* attn_buf, float* lse, const int* last, void* hs_out, void* hs_all, int M, int B, int Lmax, int cap,
                      int q_row_min, const int* dyn, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !x_in || !pos || !crow || !grow || !kv0 || !attn_buf || !lse || !workspace || M <= 0 || B <= 0) return NV_ERR_ARG;
    if ((last == nullptr) != (hs_out == nullptr)) return NV_ERR_ARG;
    if (workspace_bytes < nv_decoder_workspace_bytes(p, M)) return NV_ERR_ARG;
    const size_t d = p->d, ff = p->ff, Ms = (size_t)M;
    char* w = (char*)workspace;
    auto carve = [&](size_t bytes) { void* r = w; w += align_up(bytes); return r; };
    void* n = carve(Ms * d * 2);
    void* attn = carve(Ms * d * 2);
    void* x1 = carve(Ms * d * 2);
    void* outb[2] = {carve(Ms * d * 2), carve(Ms * d * 2)};           // layer outputs alternate between the two
    void* qkv = carve(Ms * 3 * d * 2);
    void* gu = carve(Ms * 2 * ff * 2);
    void* h = carve(Ms * ff * 2);
    float* rstd = (float*)carve(Ms * 4);
    const void* x = x_in;
    int rc = NV_OK;
#define NV_TRY(call) do { rc = (call); if (rc != NV_OK) return rc; } while (0)
    // Few rows (decode steps, pruned tails): every Linear is a weight streamer and the row kernels fold into it (nv_gemv_pre:
    // RMSNorm and the gather of the attention rows in the operand producer, SwiGLU in the gate|up epilogue; RoPE + cache scatter in one launch) --
    // 6 launches per layer instead of 11.  NV_DECODER_FUSED=0 keeps the unfused sequence (A/B, and the parity test's reference).
    const char* knob = getenv("NV_DECODER_FUSED");                   // read per call: the parity test flips it
    const char* knob2 = getenv("NV_DECODE_ATTN");
    // one new row per sample (M == B: row r is sample r): the streaming decode attention, output rows compact in `attn`
    const bool dec_attn = !(knob2 && atoi(knob2) == 0) && M == B;
    const bool fused = !(knob && atoi(knob) == 0) && M <= 16 && (d % 128) == 0 && (ff % 128) == 0;
    // weight-only fp8 with the de-quantisation overlapped (see nv_decoder_set_fp8_overlap): every layer must carry codes only
    Fp8Pipe pipe{p, (hipStream_t)stream, false, {{3 * (int)d, (int)d}, {(int)d, (int)d}, {2 * (int)ff, (int)d}, {(int)d, (int)ff}}};
    if (p->side && M > 16 && !dyn) {
        pipe.on = true;
        for (int i = 0; i < p->L && pipe.on; ++i)
            for (int k = 0; k < 4; ++k)
                if (!p->layers[i].q[k] || p->layers[i].w[k]) pipe.on = false;
    }
    if (pipe.on) {
        // the side stream starts after everything already enqueued on the caller's stream (earlier users of the panels included)
        if (hipEventRecord(p->ev_entry, pipe.main) != hipSuccess || hipStreamWaitEvent(p->side, p->ev_entry, 0) != hipSuccess)
            return NV_ERR_LAUNCH;
        NV_TRY(pipe.prefetch(0));
        NV_TRY(pipe.prefetch(1));
    }
    auto lin = [&](const Layer& ly, int i, int kind, const void* xin, void* out, const void* R, int N, int K) {
        return pipe.on ? pipe.linear(4 * i + kind, xin, out, R, M) : linear(p, ly, kind, xin, out, R, M, N, K, stream);
    };
    // Pruned tail (round 4; what LlamaStack does on the training / recompute path since round 1): the caller reads only each sample's LAST
    // row (nav_model.py:237: the <cls_1> row feeds the action head), so after the last layer's K/V have entered the cache its
    // o_proj / MLP / final norm run on B rows -- weight-streaming GEMVs -- instead of M.  NV_DECODER_PRUNE_TAIL=0: the full rows.
    static const bool prune_knob = [] { const char* e = getenv("NV_DECODER_PRUNE_TAIL"); return !e || e[0] != '0'; }();
    const bool prune_tail = prune_knob && hs_out && !hs_all && !dyn && !fused && !pipe.on && M > 16 && B <= 16;
    for (int i = 0; i < p->L; ++i) {
        const Layer& ly = p->layers[i];
        if (!ly.norm1 || !ly.norm2 || !ly.kv) return NV_ERR_ARG;
        void* x2 = outb[i & 1];                                       // != x (the previous layer wrote the other one) and != x1
        if (fused) {
            const int D = (int)d, F = (int)ff;
            auto wq = [&](int k) { return ly.q[k] ? ly.q[k] : ly.w[k]; };
            auto sq = [&](int k) { return ly.q[k] ? ly.s[k] : (const float*)nullptr; };
            NV_TRY(nv_gemv_pre(x, nullptr, wq(0), sq(0), qkv, nullptr, M, 3 * D, D, D, D, 3 * D, 0, 1, ly.norm1, p->eps, 0, stream));
            NV_TRY(nv_rope_scatter_rows_bf16(qkv, p->rope_cos, p->rope_sin, pos, crow, ly.kv, M, p->H, p->hd, 3 * D, stream));
            if (dec_attn) NV_TRY(nv_attn_decode_bf16(ly.kv, crow, pos, attn, M, p->H, p->hd, cap, stream));
            else if (dyn) NV_TRY(nv_attn_fwd_strided_dyn_bf16(ly.kv, attn_buf, lse, kv0, B, cap, p->H, p->hd, dyn, stream));
            else NV_TRY(nv_attn_fwd_strided_bf16(ly.kv, attn_buf, lse, kv0, B, Lmax, cap, p->H, p->hd, q_row_min, stream));
            NV_TRY(nv_gemv_pre(dec_attn ? attn : attn_buf, dec_attn ? nullptr : grow, wq(1), sq(1), x1, x, M, D, D, D, D, D, D, 0, nullptr, 0.f, 0,
                               stream));                                                   // x1 = x + o_proj(attn rows)
            NV_TRY(nv_gemv_pre(x1, nullptr, wq(2), sq(2), h, nullptr, M, 2 * F, D, D, D, F, 0, 1, ly.norm2, p->eps, 1, stream));   // h = SwiGLU
            NV_TRY(nv_gemv_pre(h, nullptr, wq(3), sq(3), x2, x1, M, D, F, F, F, D, D, 0, nullptr, 0.f, 0, stream));        // x2 = x1 + down(h)
            x = x2;
            continue;
        }
        NV_TRY(nv_rmsnorm_fwd_bf16(x, ly.norm1, n, rstd, M, (int)d, p->eps, stream));
        NV_TRY(lin(ly, i, 0, n, qkv, nullptr, 3 * (int)d, (int)d));
        if (knob && atoi(knob) == 0) {                                 // the literal two-launch sequence (parity reference)
            NV_TRY(nv_rope_rows_bf16(qkv, p->rope_cos, p->rope_sin, pos, M, p->H, p->hd, 3 * (int)d, stream));
            NV_TRY(nv_scatter_rows_bf16(qkv, crow, ly.kv, M, 3 * (int)d, stream));
        } else {
            NV_TRY(nv_rope_scatter_rows_bf16(qkv, p->rope_cos, p->rope_sin, pos, crow, ly.kv, M, p->H, p->hd, 3 * (int)d, stream));
        }
        if (dec_attn) {
            NV_TRY(nv_attn_decode_bf16(ly.kv, crow, pos, attn, M, p->H, p->hd, cap, stream));
        } else {
            if (dyn) NV_TRY(nv_attn_fwd_strided_dyn_bf16(ly.kv, attn_buf, lse, kv0, B, cap, p->H, p->hd, dyn, stream));
            else NV_TRY(nv_attn_fwd_strided_bf16(ly.kv, attn_buf, lse, kv0, B, Lmax, cap, p->H, p->hd, q_row_min, stream));
            NV_TRY(nv_gather_rows_bf16(attn_buf, grow, attn, M, (int)d, stream));
        }
        if (prune_tail && i == p->L - 1) {
            // the B last rows only; scratch: the q|k|v block (its rows are in the cache now) holds x_l | attn_l | x1_l
            char* q8 = (char*)qkv;
            void* x_l = q8;
            void* attn_l = q8 + align_up((size_t)B * d * 2);
            void* x1_l = q8 + 2 * align_up((size_t)B * d * 2);
            NV_TRY(nv_gather_rows_bf16(x, last, x_l, B, (int)d, stream));
            NV_TRY(nv_gather_rows_bf16(attn, last, attn_l, B, (int)d, stream));
            NV_TRY(linear(p, ly, 1, attn_l, x1_l, x_l, B, (int)d, (int)d, stream));           // M = B <= 16: the weight streamers
            NV_TRY(nv_rmsnorm_fwd_bf16(x1_l, ly.norm2, n, rstd, B, (int)d, p->eps, stream));
            NV_TRY(linear(p, ly, 2, n, gu, nullptr, B, 2 * (int)ff, (int)d, stream));
            NV_TRY(nv_swiglu_fwd_bf16(gu, h, B, (int)ff, stream));
            NV_TRY(linear(p, ly, 3, h, x2, x1_l, B, (int)d, (int)ff, stream));
            NV_TRY(nv_rmsnorm_fwd_bf16(x2, p->final_norm, hs_out, rstd, B, (int)d, p->eps, stream));
            return NV_OK;
        }
        NV_TRY(lin(ly, i, 1, attn, x1, x, (int)d, (int)d));                                  // x1 = x + o_proj(attn)
        NV_TRY(nv_rmsnorm_fwd_bf16(x1, ly.norm2, n, rstd, M, (int)d, p->eps, stream));
        NV_TRY(lin(ly, i, 2, n, gu, nullptr, 2 * (int)ff, (int)d));
        NV_TRY(nv_swiglu_fwd_bf16(gu, h, M, (int)ff, stream));
        NV_TRY(lin(ly, i, 3, h, x2, x1, (int)d, (int)ff));                                   // x2 = x1 + down(h)
        x = x2;
    }
    if (hs_all) NV_TRY(nv_rmsnorm_fwd_bf16(x, p->final_norm, hs_all, rstd, M, (int)d, p->eps, stream));
    if (hs_out) {
        NV_TRY(nv_gather_rows_bf16(x, last, n, B, (int)d, stream));
        NV_TRY(nv_rmsnorm_fwd_bf16(n, p->final_norm, hs_out, rstd, B, (int)d, p->eps, stream));
    }
#undef NV_TRY
    return NV_OK;
}

// One greedy-decoding step with every decision on the device (decode_step.hip): lm_head on the B last hidden states -> masked
// argmax + HF's finished/pad bookkeeping -> cache indices of the new token -> its embedding -> the L decoder layers over the
// K/V cache -> the new last hidden states (written back to `hs`).  Launch arguments do not change from step to step: the call
// is captured once into a hipGraph and replayed per token.
//   hs [B,d] in/out;  embed [V.., d];  lm_head [Vp, d] (rows >= V unused);  logits [B, Vp] bf16 scratch;  x [B, d] scratch;
//   state: nv_decode_state_ints(B) ints (tok|fin|len|pos|crow|grow|last|dyn|cnt);  out [max_steps, B] picked tokens
int nv_decoder_greedy_step(const nv_decoder* p, void* hs, const void* embed, const void* lm_head, int Vp, int V, int special0, int nspecial,
                           void* logits, void* x, int* state, int* out, int max_steps, const int* kv0, void* attn_buf, float* lse, int B,
                           int cap, int eos, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !hs || !embed || !lm_head || !logits || !x || !state || !out || !kv0 || !attn_buf || !lse || !workspace) return NV_ERR_ARG;
    if (B <= 0 || Vp < V || V <= 0) return NV_ERR_ARG;
    const int d = p->d;
    int rc;
    if (B <= 16 && (d & 31) == 0) rc = nv_gemv_bf16(hs, lm_head, logits, nullptr, B, Vp, d, d, d, Vp, Vp, 0, stream);
    else rc = nv_gemm_bf16_ws(0, hs, lm_head, logits, nullptr, B, Vp, d, d, d, Vp, Vp, 0, 0, p->gemm_ws, stream);
    if (rc != NV_OK) return rc;
    if ((rc = nv_decode_pick_bf16(logits, Vp, V, special0, nspecial, state, out, max_steps, B, eos, pad, stream)) != NV_OK) return rc;
    if ((rc = nv_decode_advance(state, B, cap, stream)) != NV_OK) return rc;
    int* tok = state;
    if ((rc = nv_gather_rows_bf16(embed, tok, x, B, d, stream)) != NV_OK) return rc;
    return nv_decoder_extend(p, x, state + 3 * B, state + 4 * B, state + 5 * B, kv0, attn_buf, lse, state + 6 * B, hs, nullptr, B, B, cap, cap,
                             0, state + 7 * B, workspace, workspace_bytes, stream);
}

}  // extern "C"
