/*
 * navillm_hip.h -- C ABI of libnavillm_hip.so (gfx950 / MI355X only).
 *
 * The reference (zd11024/NaviLLM) is pure Python: its drop-in boundary is the module protocol
 * `NavModel.forward(mode, batch)` (models/nav_model.py:96-126), mirrored by navillm_amd/nav_model.py.
 * Beneath that class every device computation goes through the entry points below; each one names
 * the reference call it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted "host"
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and stream-ordered
 *   - return 0 on success, <0 on error (NV_ERR_*); nothing throws, nothing allocates:
 *     scratch is caller-provided, sized by the `*_workspace_bytes` twin
 *   - bf16 tensors are raw uint16 bit patterns; rounding is round-to-nearest-even
 *   - re-entrant; no hidden global state
 */
#ifndef NAVILLM_HIP_H
#define NAVILLM_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NV_OK 0
#define NV_ERR_ARG (-1)
#define NV_ERR_SHAPE (-2)
#define NV_ERR_LAUNCH (-3)

/* ---- bf16 MFMA GEMM family: the Linear layers of HF LlamaDecoderLayer reached from
 *      models/modified_lm.py:112-116, and lm_head (modified_lm.py:120).
 *   C[M,N] = sum_k Aop[m,k]*Bop[n,k], fp32 accumulate.
 *   layout 0 "NT": A[M,K] B[N,K]  (forward  y = x W^T)          K % 64 == 0
 *   layout 1 "NN": A[M,K] B[K,N]  (dgrad    dx = dy W)           K % 64 == 0
 *   layout 2 "TN": A[K,M] B[K,N]  (wgrad    dW = dy^T x)         any K
 *   epilogue 0 store | 1 C = bf16(C + bf16(acc))  (grad accumulation)
 *            2 C = bf16(R[m,n] + bf16(acc))       (residual add)   | 3 C = bf16(acc + R[n]) (bias)
 *   tile_cfg 0 planned per shape | 1 128x128 | 8 256x256 | 84..88 the 256-wide tile cut off after 4..8 fragment rows per wave
 *   (128 / 160 / 192 / 224 / 256 x 256; forward and dgrad layouts).   lda/ldb multiples of 8, A/B 16-B aligned. */
int nv_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                 int ldb, int ldc, int ldr, int epilogue, int tile_cfg, void* stream);

/* same, with the split-K tail enabled: `workspace` = nv_gemm_bf16_workspace_bytes() bytes, zero-filled once by
 * the caller (ticket words are left zero again by the kernel), private to one stream at a time */
size_t nv_gemm_bf16_workspace_bytes(void);
int nv_gemm_bf16_ws(int layout, const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda,
                    int ldb, int ldc, int ldr, int epilogue, int tile_cfg, void* workspace, void* stream);
/*   Skinny form for the decode steps of generation (HF generate() reached from models/nav_model.py:324-341,388-402):
 *   C[M<=16,N] = A[M,K] @ W[N,K]^T (+ R); a weight streamer (HBM-bound, N*K*2 bytes per launch); epilogue 0 = store,
 *   2 = residual add with nv_gemm_bf16's rounding order; K % 32 == 0. */
int nv_gemv_bf16(const void* A, const void* W, void* C, const void* R, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
                 int epilogue, void* stream);
/*   packed q|k|v projection with RoPE (modified_lm.py:112-116 -> HF LlamaAttention) applied to the first rope_cols
 *   columns in the GEMM epilogue: position of row m = pos[m] (int32, packed rows) or, with pos == NULL, m % S; tables as
 *   nv_rope_bf16. Bit-identical to
 *   nv_gemm_bf16(NT) + nv_rope_bf16. */
int nv_gemm_bf16_rope(const void* A, const void* W, void* C, const void* rope_cos, const void* rope_sin, const int* pos, int M,
                      int N, int K, int lda, int ldw, int ldc, int S, int rope_cols, void* workspace, void* stream);
/*   the same with an explicit tile configuration (nv_gemm_bf16_ws's `tile_cfg`: 0 planned, 8 = 256x256, 84..88 = the 256-wide tile
 *   cut off after 4..8 fragment rows per wave) */
int nv_gemm_bf16_rope_cfg(const void* A, const void* W, void* C, const void* rope_cos, const void* rope_sin, const int* pos, int M,
                          int N, int K, int lda, int ldw, int ldc, int S, int rope_cols, int tile_cfg, void* workspace, void* stream);

/* ---- weight-only fp8 (OCP e4m3fn, one fp32 scale per output channel) for the LM's Linear layers: SURVEY.md §8f item 4 /
 *      BASELINE config 5 (Vicuna-13B inference).  W[n,:] ~= s[n]*q[n,:], s[n] = max|W[n,:]|/448, q = e4m3fn(W/s) (RNE).  The
 *      arithmetic everywhere is the bf16 GEMM on the de-quantised weight bf16(s*q) ("the reference run on de-quantised weights").
 *   quant: bf16 [N,K] -> codes u8 [N,K] + scales [N]  (K % 8 == 0)    dequant: -> bf16 [N,K]  (K % 16 == 0) */
int nv_fp8_quant_rows(const void* W, void* Q, float* scales, int N, int K, int ldw, int ldq, void* stream);
int nv_fp8_dequant_rows(const void* Q, const float* scales, void* out, int N, int K, int ldq, int ldo, void* stream);
/*   the in-register decode of all 256 codes (bf16 out[256]); pinned against torch.float8_e4m3fn by the tests */
int nv_fp8_decode_table(void* out256_bf16, void* stream);
/*   decode-step weight streamer on fp8 weights: C[M<=16,N] = A[M,K] @ bf16(s*q)^T (+R); K % 64 == 0; epilogue 0 | 2 as nv_gemv_bf16 */
int nv_gemv_fp8w(const void* A, const void* Wq, const float* scales, void* C, const void* R, int M, int N, int K, int lda, int ldw,
                 int ldc, int ldr, int epilogue, void* stream);

/*   the tile GEMM on the CODES themselves (round 4; "fp8 MFMA path" of BASELINE config 5): C[M,N] = A[M,K] @ bf16(s*q)^T (+R) with the
 *   weight tile DMA'd as e4m3fn bytes (half the fabric / LDS bytes of the bf16 operand) and converted on the MFMA fragment path.
 *   Serves the few-hundred-row GEMMs of K/V-reuse steps -- the shapes whose launch plan is a 128 / 160-row cut-off tile; every
 *   other shape returns NV_ERR_SHAPE (-2) and the caller runs nv_fp8_dequant_rows + nv_gemm_bf16.  mode: 0 = default, 7 = operand
 *   bf16(s*q) bit for bit as nv_fp8_dequant_rows writes it, 9 = v_cvt_scalef32_pk_bf16_fp8 unscaled + s[n] on the fp32 accumulator
 *   (one bf16 rounding per weight less than the reference-on-de-quantised-weights semantics); 8 = the same instruction with s as its
 *   scale operand: a measurement only -- the hardware uses the operand's exponent alone, the results are WRONG by up to 2x.  tile_cfg 0 | 84 | 85; epilogue 0 (store) | 2 (residual); K % 64 == 0, ldq % 16 == 0; workspace as nv_gemm_bf16_ws */
int nv_gemm_fp8w(const void* A, const void* codes, const float* scales, void* C, const void* R, int M, int N, int K, int lda, int ldq,
                 int ldc, int ldr, int epilogue, int mode, int tile_cfg, void* workspace, void* stream);
/*   process-wide default for mode = 0 (7 | 9; 0 = query only); returns the previous default */
int nv_gemm_fp8w_default_mode(int mode);

/* ---- K6: embedding gather + visual-token add, models/modified_lm.py:100-110.
 *   out[m] = table[ids[m]]  or  bf16(f32(table[ids[m]]) + vis[vis_idx[m]])  when vis_idx[m] >= 0 */
int nv_embed_vis_bf16(const void* table, const int* ids, const int* vis_idx, const float* vis, void* out, int M, int d,
                      void* stream);
/*   backward of the add: dvis[i] = f32(dE[vis_rows[i]]) */
int nv_vis_grad_f32(const void* dE, const int* vis_rows, float* dvis, int nvis, int d, void* stream);
/*   embedding-table gradient; tokens grouped by id on the host: segment u = tok[seg_off[u]..seg_off[u+1]) */
int nv_embed_grad_bf16(const void* dE, const int* uniq, const int* seg_off, const int* tok, void* gtable, int n_uniq, int d,
                       void* stream);

/* ---- K7c: HF LlamaRMSNorm (fp32 statistics), forward and backward (+fused residual-grad add,
 *      weight grad accumulated into gw as bf16(gw + bf16(sum))) */
int nv_rmsnorm_fwd_bf16(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps, void* stream);
size_t nv_rmsnorm_bwd_workspace_bytes(int d);
int nv_rmsnorm_bwd_bf16(const void* dy, const void* x, const void* w, const float* rstd, const void* resid_grad, void* dx,
                        void* gw, void* workspace, int M, int d, void* stream);

/* ---- HF apply_rotary_pos_emb on the q and k parts of packed qkv [M, 3*H*hd] in place; position of
 *      row m is m % S (arange(S) incl. left padding); tables [maxS, hd] bf16 as HF builds them;
 *      backward != 0 applies the transpose */
int nv_rope_bf16(void* qkv, const void* cos_t, const void* sin_t, int M, int S, int H, int hd, int ld, int backward,
                 void* stream);
/*   per-row positions pos[m] (forward only): the position_ids of HF's generation path (left-aligned per-sample frames) */
int nv_rope_rows_bf16(void* qkv, const void* cos_t, const void* sin_t, const int* pos, int M, int H, int hd, int ld, void* stream);

/*   transpose of nv_rope_rows_bf16 (backward of the per-row-position RoPE) */
int nv_rope_rows_t_bf16(void* qkv, const void* cos_t, const void* sin_t, const int* pos, int M, int H, int hd, int ld, void* stream);
/*   K/V gradients of a cached prompt prefix, summed over the steps of an episode in fp32 (navillm_amd/episode.py):
 *   accum : acc[rows[i], 0..2d) += dqkv[rows[i], d..3d)        inject: dqkv[i, d..3d) += acc[rows[i], 0..2d)  (dqkv bf16 [*,3d]) */
int nv_kv_grad_accum_f32(const void* dqkv, float* acc, const int* rows, int n, int d, void* stream);
/*   set   : acc[rows[i], 0..2d)  = dqkv[rows[i], d..3d)        (the episode's first step: the accumulator is never zero-filled) */
int nv_kv_grad_set_f32(const void* dqkv, float* acc, const int* rows, int n, int d, void* stream);
int nv_kv_grad_inject_bf16(void* dqkv, const float* acc, const int* rows, int n, int d, void* stream);

/* ---- HF LlamaMLP activation on packed gate|up [M, 2*ff]: h = bf16(bf16(silu(g)) * u) */
int nv_swiglu_fwd_bf16(const void* gu, void* h, int M, int ff, void* stream);
int nv_swiglu_bwd_bf16(const void* gu, const void* dh, void* dgu, int M, int ff, void* stream);

int nv_scale_bf16(const void* x, void* out, long n, float scale, void* stream);   /* n % 8 == 0 */
/*   scale read from DEVICE memory; accumulate != 0: out = bf16(out + bf16(x*scale)) (the LM-loss coefficient applied to the
 *   lm_head gradients without a host round trip, models/modified_lm.py:126-137 + mp3d_agent.py:865,901) */
int nv_scale_dev_bf16(const void* x, void* out, long n, const float* scale_dev, int accumulate, void* stream);
int nv_gather_rows_bf16(const void* src, const int* rows, void* out, int n, int d, void* stream);
int nv_scatter_rows_bf16(const void* src, const int* rows, void* dst, int n, int d, void* stream);

/* ---- K7b: causal + left-pad attention of HF LlamaAttention (head_dim 128), flash style.
 *   qkv [B*S, 3*H*128] post-RoPE, out [B*S, H*128], lse2 [B,H,S] (log2 domain, +inf for fully
 *   masked rows), kv_start[b] = number of left-pad positions of sample b. */
/*   q_row_min (multiple of 128, 0 = all): only query rows >= q_row_min of every sample are computed (forward) or
 *   carry gradient (backward; dQ rows below it are left untouched) -- the navigation modes read one row. */
int nv_attn_fwd_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S, int H, int head_dim,
                     int q_row_min, void* stream);
/*   KV-cache layout (inference; SURVEY.md §8f items 1-2, HF generation path reached from models/modified_lm.py:184-199):
 *   sample b's rows start at b*S_stride in qkv, out and lse2; S = longest valid length. */
/*   ... with the cache length and the first computed query row in DEVICE memory: dyn = {S, q_row_min} (a decode step replayed
 *   from a hipGraph has frozen launch arguments); the grid covers S_stride, query blocks at or beyond S exit at once */
/*   decode attention (one new token per sample): query r = q slice of cache row crow[r] of sample r, keys/values = cache rows
 *   r*cap + [0, pos[r]]; out [M, H*128] compact; HBM-bound streaming form, K and V read once */
int nv_attn_decode_bf16(const void* kv, const int* crow, const int* pos, void* out, int M, int H, int head_dim, int cap, void* stream);
int nv_attn_fwd_strided_dyn_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S_stride, int H, int head_dim,
                                 const int* dyn, void* stream);
int nv_attn_fwd_strided_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, int B, int S, int S_stride, int H,
                             int head_dim, int q_row_min, void* stream);
/*   Packed ("varlen") rows: sample b = rows [cu[b], cu[b+1]) (cu: int32 [B+1], device), S_max = longest sample, lse2
 *   [B,H,S_max]; no padding keys; pos0[b] = position of the sample's first token (the reference numbers positions over
 *   its left padding) -- read only by the backward's fused RoPE^T.  q_row_min >= 0 as above, -1 =
 *   each sample's own last 128-row block.  The LM then never computes the left-padding rows of a batch. */
int nv_attn_fwd_varlen_bf16(const void* qkv, void* out, float* lse2, const int* cu, const int* pos0, int B, int S_max, int H,
                            int head_dim, int q_row_min, void* stream);
/*   PARITY INSTRUMENT (tests only, not on the product path): the forward with the rounding points of HF's eager
 *   LlamaAttention in bf16 -- scores = bf16(bf16(q k^T) * hd^-0.5), P = bf16(softmax_fp32(scores)), out = bf16(P v) -- computed in
 *   two passes.  cu == NULL: padded layout (as nv_attn_fwd_bf16), else packed rows (as nv_attn_fwd_varlen_bf16, kv_start = pos0).
 *   Used to show that the distance between the product kernel (fp32 scores, flash style) and the reference's bf16 run is this
 *   difference in rounding points and nothing else. */
int nv_attn_fwd_hfround_bf16(const void* qkv, void* out, float* lse2, const int* kv_start, const int* cu, int B, int S, int H,
                             int head_dim, int q_row_min, void* stream);
/*   backward over the K/V-cache layout of nv_attn_fwd_strided_bf16 (training with a cached prompt prefix, navillm_amd/episode.py):
 *   queries >= q_row_min carry gradient (dO of all other rows zero), dK/dV for every key row < S, no RoPE^T;
 *   workspace = nv_attn_bwd_workspace_bytes(B, S_stride, H) */
int nv_attn_bwd_strided_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                             void* workspace, int B, int S, int S_stride, int H, int head_dim, int q_row_min, void* stream);
/*   the same with the K/V gradients of each sample's first prefix_len[b] key rows accumulated in fp32 into kv_acc [B*S_stride,
 *   2*H*head_dim] straight from the MFMA accumulators (first != 0: stored) instead of written to dqkv -- the fused form of
 *   nv_kv_grad_accum_f32 / nv_kv_grad_set_f32 */
int nv_attn_bwd_strided_kvacc_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                                   void* workspace, float* kv_acc, const int* prefix_len, int first, int B, int S, int S_stride, int H,
                                   int head_dim, int q_row_min, void* stream);
/*   attention backward of ALL T steps of a prefix-reuse episode for one layer, in place on the episode's row buffers
 *   (navillm_amd/episode.py, mode "all"; the reference runs one full-prompt backward per step, mp3d_agent.py:756): rows [0, Mp) =
 *   the packed prompt prefixes (cu [B+1]), then T packed step blocks described by tab (device int32: off[T*B] = first row of step t,
 *   sample b | n[T*B] = its rows; together they tile [Mp, R)); lse_ptrs = T device pointers to the steps' lse2 [B, H, cap] (indexed by
 *   cache position prefix_len + j).  Writes dqkv rows [Mp, R) (dQ and the steps' own dK|dV, through RoPE^T when the tables are given,
 *   position = prefix_len + j) and STORES the fp32 sum over the steps of the prefix rows' dK|dV in kv_acc [B*cap, 2*H*head_dim]
 *   (row b*cap + key).  workspace: (R - Mp) * H floats; at most 128 steps, at most 1536 rows per sample and step */
/*   round 5: the FORWARD attention of all T steps of a prefix-reuse episode in one launch, reading the episode row buffers in place
 *   (no scatter into the K/V-cache layout and back): qkv [rows, 3*H*128] post-RoPE -- prefix rows of sample b at [cu[b], cu[b+1]), step
 *   t's rows of sample b at [tab[t*B+b], + tab[T*B+t*B+b]) --, out [rows, H*128] written at the steps' rows, lse_ptrs = device array of T
 *   pointers to fp32 [B, H, cap] (step t's log2-domain lse at cache position prefix_len + j).  n_max = longest per-sample row count of a
 *   step.  Bit-identical to scatter + nv_attn_fwd_strided_bf16 + gather per step. */
int nv_attn_fwd_episode_bf16(const void* qkv, void* out, const void* lse_ptrs, const int* cu, const int* tab, int T, int B, int H,
                             int head_dim, int cap, int n_max, long rows, void* stream);
int nv_attn_bwd_episode_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, void* workspace, const void* lse_ptrs,
                             const int* cu, const int* tab, float* kv_acc, const void* rope_cos, const void* rope_sin, int T, int B, int H,
                             int head_dim, int cap, int Mp, long R, int Lp_max, int N_max, void* stream);
/*   the same; accumulate != 0 ADDS the prefix rows' dK|dV to kv_acc instead of storing them: later segments of a long episode whose
 *   deferred backward is flushed in pieces (navillm_amd/episode.py; the first segment stores, the following ones add) */
int nv_attn_bwd_episode_acc_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, void* workspace, const void* lse_ptrs,
                                 const int* cu, const int* tab, float* kv_acc, const void* rope_cos, const void* rope_sin, int T, int B, int H,
                                 int head_dim, int cap, int Mp, long R, int Lp_max, int N_max, int accumulate, void* stream);
int nv_attn_bwd_varlen_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* cu, const int* pos0,
                            void* dqkv, void* workspace, const void* rope_cos, const void* rope_sin, int B, int S_max, long rows,
                            int H, int head_dim, int q_row_min, void* stream);
size_t nv_attn_bwd_workspace_bytes(int B, int S, int H);
int nv_attn_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                     void* workspace, int B, int S, int H, int head_dim, int q_row_min, void* stream);
/*   the same, with the transpose of RoPE applied to dQ and dK as they are written (bit-identical to nv_attn_bwd_bf16
 *   followed by nv_rope_bf16(backward=1) on dqkv; rope_cos/rope_sin = that function's tables) */
int nv_attn_bwd_rope_bf16(const void* qkv, const void* out, const void* dout, const float* lse2, const int* kv_start, void* dqkv,
                          void* workspace, const void* rope_cos, const void* rope_sin, int B, int S, int H, int head_dim,
                          int q_row_min, void* stream);

/* ---- K10: action / object head Linear(d -> N<=128) in the LM dtype, models/nav_model.py:237,445 */
int nv_head_fwd_bf16(const void* x, const void* W, const void* bias, void* y, int B, int d, int N, void* stream);
int nv_head_bwd_bf16(const void* dy, const void* x, const void* W, void* dx, void* gW, void* gb, int B, int d, int N,
                     void* stream);

/* ---- K11: CrossEntropyLoss(ignore_index=-100, reduction='sum') on bf16 logits with -inf slots,
 *      train.py:229 / tasks/agents/mp3d_agent.py:750. loss_rows[b] fp32; dlogits = bf16(gscale*(p-onehot)) */
int nv_action_ce_bf16(const void* logits, const long* targets, float* loss_rows, void* dlogits, int B, int G, float gscale,
                      const float* gscale_dev /* optional device scalar multiplied into gscale */, void* stream);
/* ---- K9: token CE of models/modified_lm.py:122-137 on materialised logits (special ids = -inf,
 *      labels already shifted on the host, -100 ignored); logits overwritten by their gradient */
int nv_lm_ce_bf16(void* logits, const int* labels, float* loss_rows, int M, int V, int ldl, int special0, int nspecial,
                  float gscale, int write_grad, void* stream);

/* ---- K13: clip_grad_norm_(40) + AdamW on flat parameter buffers, train.py:86-89, tools/optims.py:43-45 */
int nv_sumsq(const void* g, long n, int is_bf16, float* partial, int* n_partial_host, void* stream);
int nv_clip_coef(const float* partial, int n_partial, float max_norm, float* out2, void* stream);
int nv_adamw(void* p, const void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
             double wd, int step, const float* clip_out2, void* stream);
/*   the same with `optimizer.zero_grad()` (train.py:88-89) folded in: g is zeroed as it is consumed -- no separate fill pass over the
 *   gradient buffer */
int nv_adamw_zero_grad(void* p, void* g, void* m, void* v, long n, int is_bf16, double lr, double beta1, double beta2, double eps,
                       double wd, int step, const float* clip, void* stream);

/* ---- fp32 scene encoder + fusion (K1-K5, K12): models/image_embedding.py:51-121,
 *      models/detr_transformer.py:170-182, models/nav_model.py:146-194 */
int nv_gemm_f32(int layout, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                int ldc, int accumulate, void* stream);
/*   with a caller workspace: launches with few output tiles and a long K (the encoder's [288 x 1024] results) are split
 *   into up to 8 K slices summed in slice order (deterministic) */
size_t nv_gemm_f32_workspace_bytes(int M, int N);
int nv_gemm_f32_ws(int layout, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                   int ldc, int accumulate, void* workspace, void* stream);
int nv_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int M, int d,
                         float eps, void* stream);
size_t nv_layernorm_bwd_workspace_bytes(int d);
int nv_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                         float* gw, float* gb, void* workspace, int M, int d, int accumulate, void* stream);
int nv_colsum_f32(const float* x, float* out, int M, int d, int ld, int accumulate, void* stream);
int nv_mha_fwd_f32(const float* qkv, const int* lens, float* out, float* P, int B, int N, int heads, int hd, void* stream);
int nv_mha_bwd_f32(const float* qkv, const float* P, const float* dout, float* dqkv, int B, int N, int heads, int hd,
                   void* stream);
/*   the same with nn.MultiheadAttention(dropout=p)'s attention-probability dropout (detr_transformer.py:138; torch drops the
 *   softmax output before P.V and rescales by 1/(1-p)).  keep (optional): [B,heads,N,N] 0/1 flags injected by parity tests;
 *   otherwise element e of P keeps iff Philox4x32-10(counter = offset + e, key = seed) word 0 >= p.  P receives the
 *   probabilities BEFORE dropout; the backward regenerates the mask from the same (keep, p, seed, offset). */
int nv_mha_fwd_drop_f32(const float* qkv, const int* lens, float* out, float* P, const float* keep, float p,
                        unsigned long long seed, unsigned long long offset, int B, int N, int heads, int hd, void* stream);
int nv_mha_bwd_drop_f32(const float* qkv, const float* P, const float* dout, float* dqkv, const float* keep, float p,
                        unsigned long long seed, unsigned long long offset, int B, int N, int heads, int hd, void* stream);
int nv_gelu_fwd_f32(const float* x, float* y, long n, void* stream);
int nv_gelu_bwd_f32(const float* x, const float* dy, float* dx, long n, void* stream);
int nv_add_f32(const float* a, const float* b, float* out, long n, int d, int b_bcast, void* stream);
int nv_mul_f32(const float* a, const float* b, float* out, long n, void* stream);
/*   nn.Dropout(p) with an in-kernel Philox4x32-10 mask: out = keep ? x/(1-p) : 0, keep(i) from (seed, offset + i/4).  The
 *   backward is the same call on the gradient with the same (seed, offset): the mask is regenerated, never stored. */
int nv_dropout_f32(const float* x, float* out, long n, float p, unsigned long long seed, unsigned long long offset, void* stream);
int nv_rowscale_f32(const float* x, const float* s, float* out, long rows, int d, void* stream);
int nv_gather_add_f32(const float* src, const int* idx, const float* base, float* out, long rows, int d, void* stream);
int nv_index_sum_f32(const float* src, const int* idx, float* dst, int n, int R, int d, int accumulate, void* stream);
int nv_masked_mean_f32(const float* x, const float* mask, float* out, int B, int N, int d, void* stream);

/* ---- HOST side-car of the navigation step (pure host code, no device pointers): the topological map of
 *      models/graph_utils.py:47-165 on dense matrices with integer node ids, and the per-step index tables of
 *      models/nav_model.py:174-190,216-223,234-242.  Output buffers are plain host memory (pinned by the caller so the
 *      following H2D copies are asynchronous). */
typedef struct nv_graph nv_graph;
nv_graph* nv_graph_create(void);
void nv_graph_destroy(nv_graph* g);
int nv_graph_add_node(nv_graph* g);                                   /* -> new node id 0,1,2,... */
int nv_graph_num_nodes(const nv_graph* g);
int nv_graph_set_position(nv_graph* g, int node, const double* xyz);
int nv_graph_add_edge(nv_graph* g, int x, int y, double dist);        /* FloydGraph.add_edge, graph_utils.py:59-64 */
int nv_graph_update(nv_graph* g, int k);                              /* FloydGraph.update,   graph_utils.py:66-75 */
int nv_graph_visited(const nv_graph* g, int k);
double nv_graph_distance(const nv_graph* g, int x, int y);            /* 95959595 = no path */
int nv_graph_path(const nv_graph* g, int x, int y, int* out, int cap);/* FloydGraph.path: -> length, ids [v1..y] */
/*   GraphMap.get_pos_fts (graph_utils.py:144-165) for n slots at once (ids[i] < 0 = the `None` slot): out [n, angle_feat_size+3] */
int nv_graph_pos_fts(const nv_graph* g, int cur, const int* ids, int n, double cur_heading, double cur_elevation,
                     int angle_feat_size, float* out);
/*   GraphMap.node_step_ids[vp] = step (tasks/agents/mp3d_agent.py:688) */
int nv_graph_set_step_id(nv_graph* g, int node, int step);
/*   MP3DAgent.nav_gmap_variable + the pose half of nav_vp_variable (tasks/agents/mp3d_agent.py:264-371) for a whole batch in one
 *   call: map slots [stop] + visited + unvisited nodes per sample, padded to the batch's longest list G (returned; <= Gcap, the
 *   row capacity of every buffer); gmap_ids [B,G] i32, gmap_step_ids [B,G] i64, gmap_visited / gmap_masks [B,G] u8,
 *   gmap_pos_fts [B,G,afs+3] f32, gmap_lens [B], no_vp_left [B] u8, pair_dists [B,G,G] f32 or NULL, vp_pos_fts [B,Nv,2(afs+3)] f32,
 *   vp_cand_ids [B,Nv] i32.  cand_ids = the panoramas' candidate node ids, sample b's at cand_off[b]..cand_off[b+1]. */
int nv_nav_collate(nv_graph* const* graphs, int B, const int* cur, const int* start, const double* heading, const double* elevation,
                   const int* cand_ids, const int* cand_off, int Nv, int angle_feat_size, int enc_full_graph, int Gcap,
                   int* gmap_ids, long* gmap_step_ids, unsigned char* gmap_visited, unsigned char* gmap_masks, float* gmap_pos_fts,
                   int* gmap_lens, unsigned char* no_vp_left, float* pair_dists, float* vp_pos_fts, int* vp_cand_ids);
/*   nav_model.py:174-190: src [B*G] (view row feeding each map slot or -1), inv [B*Nv] (its inverse), ttype [B*G] */
int nv_nav_match_tables(const int* gmap_ids, const unsigned char* gmap_visited, const int* cand_ids, int B, int G, int Nv, int* src,
                        int* inv, int* ttype);
/*   nav_model.py:216-223,234-242: candidate rows in LM order under the per-sample permutations, its inverse, and the head
 *   column of every map slot; returns the number of selected rows */
int nv_nav_perm_tables(const unsigned char* cand_mask, const long* perm, const int* perm_off, int B, int G, int* sel, int* inv_sel,
                       long* col);

/* ---- the inference-time decoder stack as one native call (navillm_amd/csrc/decoder_runtime.cpp): a
 *      table of layer pointers (bf16 operands or weight-only fp8 codes + scales, norms, the layer's K/V cache) and one entry
 *      point that enqueues every launch of a K/V-reuse / decode step on `stream`, intermediates carved from `workspace`.
 *      Counterpart: HF LlamaModel.forward with past_key_values, reached from models/modified_lm.py:184-199. */
typedef struct nv_decoder nv_decoder;
nv_decoder* nv_decoder_create(int L, int d, int H, int head_dim, int ff, float eps);
void nv_decoder_destroy(nv_decoder* p);
/*   kind: 0 q|k|v [3d,d], 1 o [d,d], 2 gate|up [2ff,d], 3 down [d,ff]; either w (bf16) or codes + scales (e4m3fn, per output channel) */
int nv_decoder_set_weight(nv_decoder* p, int layer, int kind, const void* w, const void* codes, const float* scales);
int nv_decoder_set_layer(nv_decoder* p, int layer, const void* input_norm, const void* post_attn_norm, void* kv_cache);
int nv_decoder_set_shared(nv_decoder* p, const void* rope_cos, const void* rope_sin, const void* final_norm, void* gemm_workspace,
                          void* fp8_scratch);
/*   weight-only fp8 without resident bf16 operands: two bf16 panels (each >= the largest operand) + a side stream; nv_decoder_extend
 *   then de-quantises the NEXT Linear's operand on the side stream while the current GEMM runs (steps of > 16 rows) */
int nv_decoder_set_fp8_overlap(nv_decoder* p, void* panel_a, void* panel_b, void* side_stream);
/*   nv_gemm_fp8w mode of THIS decoder's few-hundred-row GEMMs on fp8 codes: 7 | 9, 0 = the process default */
int nv_decoder_set_fp8_gemm_mode(nv_decoder* p, int mode);
size_t nv_decoder_workspace_bytes(const nv_decoder* p, int max_rows);
/*   x_in [M,d] new-row embeddings; pos/crow/grow [M] (position, cache row written, cache row read back); kv0 [B] zeros; attn_buf
 *   [B*cap,d]; lse [B,H,cap]; last [B] -> hs_out [B,d] final-norm hidden states of those block rows; hs_all optional [M,d] */
/*   dyn: optional DEVICE {Lmax, q_row_min} overriding the two host values */
int nv_decoder_extend(const nv_decoder* p, const void* x_in, const int* pos, const int* crow, const int* grow, const int* kv0,
                      void* attn_buf, float* lse, const int* last, void* hs_out, void* hs_all, int M, int B, int Lmax, int cap,
                      int q_row_min, const int* dyn, void* workspace, size_t workspace_bytes, void* stream);
/* ---- the decode-step Linear with its row kernels folded in (navillm_amd/csrc/gemv_stream.hip; HF LlamaDecoderLayer's
 *      RMSNorm -> Linear and gate|up -> SwiGLU as one launch each when M <= 16):  C = pre(A)[M,K] @ W[N,K]^T
 *      W bf16 (scales NULL) or e4m3fn codes + per-row scales; a_rows: optional row gather of A;
 *      rmsnorm: the operand is RMSNorm(A; norm_w [K], eps), bit-identical to nv_rmsnorm_fwd_bf16;
 *      swiglu == 0: C [M,N] (+ optional residual R);  swiglu != 0 (with rmsnorm, no R): W = gate|up [N = 2 ff, K] and
 *      C [M, N/2] = bf16(bf16(silu(g)) * u) of the bf16-rounded projections, bit-identical to nv_swiglu_fwd_bf16 on the stored gate|up.
 *      NV_ERR_SHAPE outside the streamer's fast path (K % 64, fp8: % 128; N % 8; 16-B aligned rows; M <= 16). */
int nv_gemv_pre(const void* A, const int* a_rows, const void* W, const float* scales, void* C, const void* R, int M, int N, int K, int lda,
                int ldw, int ldc, int ldr, int rmsnorm, const void* norm_w, float eps, int swiglu, void* stream);
/*   nv_rope_rows_bf16 + nv_scatter_rows_bf16 in one launch: dst[rows[m]] = [rope(q) | rope(k) | v] of qkv row m at position pos[m] */
int nv_rope_scatter_rows_bf16(const void* qkv, const void* cos_t, const void* sin_t, const int* pos, const int* rows, void* dst, int M, int H,
                              int hd, int ld, void* stream);
/* ---- greedy decoding with the decisions on the device (HF generate(do_sample=False), models/nav_model.py:324-341,388-402;
 *      special-id mask models/modified_lm.py:122-124).  state = nv_decode_state_ints(B) int32:
 *      tok[B] | fin[B] | len[B] | pos[B] | crow[B] | grow[B] | last[B] | dyn[2] | cnt[1] | overflow[1].  `overflow` is sticky: set when a
 *      sample's cache is full (len == cap) at a step -- that sample is marked finished and its further tokens are `pad`; the caller
 *      must treat a non-zero overflow word as an error (state[7*B+3]).
 *      nv_decode_pick_bf16: masked argmax of logits [B, ldl] (ids >= V and [special0, special0+nspecial) excluded, ties -> smallest
 *      id), finished rows emit `pad`, a row finishes on `eos`; out[cnt, b] = token.  nv_decode_advance: cache indices of the new
 *      token from len, dyn = {max len + 1, 128-aligned min len}, len += 1, cnt += 1.  nv_decoder_greedy_step: lm_head -> pick ->
 *      advance -> embedding gather -> nv_decoder_extend(dyn); identical launch arguments every step (hipGraph replay). */
int nv_decode_state_ints(int B);
int nv_decode_pick_bf16(const void* logits, int ldl, int V, int special0, int nspecial, int* state, int* out, int max_steps, int B, int eos,
                        int pad, void* stream);
int nv_decode_advance(int* state, int B, int cap, void* stream);
int nv_decoder_greedy_step(const nv_decoder* p, void* hs, const void* embed, const void* lm_head, int Vp, int V, int special0, int nspecial,
                           void* logits, void* x, int* state, int* out, int max_steps, const int* kv0, void* attn_buf, float* lse, int B,
                           int cap, int eos, int pad, void* workspace, size_t workspace_bytes, void* stream);

/* ---- data-parallel exchange over RCCL (C0-C3): replaces the DDP gradient all-reduce behind tools/optims.py:52-54
 *      (+ its initial parameter broadcast) and the task-id broadcast of tasks/loaders.py:176-179.
 *      One communicator per process/GPU; RCCL is bound at run time (the copy torch loaded, else /opt/rocm's).
 *      Collectives are in place, stream-ordered, never allocate.  Returns NV_OK, NV_ERR_ARG (-1) or -4 (RCCL failure
 *      / librccl not found). */
typedef struct nv_ctx nv_ctx;
int nv_comm_unique_id_bytes(void);                       /* 128 */
int nv_comm_unique_id(void* id_out);                     /* rank 0; ship the bytes to the other ranks out of band */
int nv_comm_rccl_version(void);                          /* NCCL_VERSION_CODE of the librccl bound at run time (22707 on ROCm 7.2) */
/*   nv_comm_init refuses a librccl whose major version is not 2 and then VERIFIES the enum values this library restates from
 *   rccl.h (ncclAvg, ncclBfloat16, ncclFloat32) with a mean all-reduce of known bf16 / fp32 vectors over the new communicator */
int nv_comm_init(nv_ctx** out, const void* id, int rank, int world);   /* collective; current HIP device */
int nv_comm_rank(const nv_ctx* c);
int nv_comm_world(const nv_ctx* c);
int nv_comm_allreduce_bf16(nv_ctx* c, void* buf, long count, int average, void* stream);
int nv_comm_allreduce_f32(nv_ctx* c, void* buf, long count, int average, void* stream);
/*   the same mean in two phases for the fully connected xGMI node: in-place reduce-scatter (rank r keeps
 *   the reduced chunk r of `count`/world elements) + in-place all-gather of the chunks; count resp. bytes % world == 0 */
int nv_comm_reduce_scatter(nv_ctx* c, void* buf, long count, int is_bf16, int average, void* stream);
int nv_comm_all_gather(nv_ctx* c, void* buf, long bytes, void* stream);
int nv_comm_broadcast(nv_ctx* c, void* buf, long bytes, int root, void* stream);
int nv_comm_destroy(nv_ctx* c);

#ifdef __cplusplus
}
#endif
#endif
