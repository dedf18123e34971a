"""Raw (non-autograd) tensor-level wrappers over the C ABI of libnavillm_hip.so.

torch is used here only for device memory (tensors), the current HIP stream and shapes.
Every function launches hand-written gfx950 kernels; nothing falls back to torch math.
"""
import ctypes
import torch
from . import debug as _debug
from . import lib as _lib

_debug.install_if_enabled()          # NAVILLM_POISON=1: NaN-filled, canaried allocations (navillm_amd/debug.py)

BF16 = torch.bfloat16
F32 = torch.float32
EPI_STORE, EPI_ACCUM, EPI_RESID, EPI_BIAS, EPI_SWIGLU_BWD = 0, 1, 2, 3, 4
NT, NN, TN = 0, 1, 2


def _L():
    return _lib.load()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_idx = None


def _st():
    """raw hipStream_t of torch's current stream.  `torch.cuda.current_stream()` costs ~13 us of Python per call (it was
    a third of the host time of a cached-inference step); the C accessor is ~0.2 us.  One process drives one GPU
    (parallel.py), so the device index is resolved once."""
    global _dev_idx
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev_idx is None:
        _dev_idx = torch.cuda.current_device()
    return _raw_stream(_dev_idx)


def _p(t):
    return 0 if t is None else t.data_ptr()


def h2d(t, device, dtype=None):
    """host -> device copy that does not stall the host: a pageable-memory copy blocks until the stream has
    drained (i.e. until the previous step's backward is done), which serialises the host-side input building
    with the GPU.  Staging through torch's cached pinned pool makes the copy asynchronous and stream-ordered."""
    t = torch.as_tensor(t)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    device = torch.device(device)
    if t.device.type == "cpu" and device.type == "cuda":
        return t.contiguous().pin_memory().to(device, non_blocking=True)
    return t.to(device)


def _chk2d(t, dt):
    assert t.is_cuda and t.dtype == dt and t.dim() == 2 and t.stride(1) == 1, (t.shape, t.dtype, t.stride())


# ------------------------------------------------------------------ bf16 GEMM
def gemm_bf16(layout, A, B, out=None, R=None, epilogue=EPI_STORE, tile_cfg=0):
    """C = Aop @ Bop^T (see include/navillm_hip.h). A, B, out, R: 2-D bf16, unit inner stride."""
    _chk2d(A, BF16)
    _chk2d(B, BF16)
    if layout == NT:
        M, K = A.shape
        N, K2 = B.shape
    elif layout == NN:
        M, K = A.shape
        K2, N = B.shape
    else:
        K, M = A.shape
        K2, N = B.shape
    assert K == K2, (A.shape, B.shape, layout)
    wide = 2 if epilogue == EPI_SWIGLU_BWD else 1           # SwiGLU-backward epilogue: out = d(gate|up) [M, 2N], R = gate|up
    if out is None:
        out = torch.empty((M, wide * N), dtype=BF16, device=A.device)
    _chk2d(out, BF16)
    assert tuple(out.shape) == (M, wide * N)
    ldr = 0
    if epilogue in (EPI_RESID, EPI_SWIGLU_BWD):
        _chk2d(R, BF16)
        assert tuple(R.shape) == (M, wide * N)
        ldr = R.stride(0)
    if GEMV_DECODE and layout == NT and M <= 16 and epilogue in (EPI_STORE, EPI_RESID) and K % 32 == 0 and tile_cfg == 0:
        # decode step of generation: a handful of token rows against a whole weight matrix -> weight-streaming kernel
        rc = _L().nv_gemv_bf16(A.data_ptr(), B.data_ptr(), out.data_ptr(), _p(R), M, N, K, A.stride(0), B.stride(0), out.stride(0),
                               ldr, epilogue, _st())
        _lib.check(rc, "nv_gemv_bf16")
        return out
    rc = _L().nv_gemm_bf16_ws(layout, A.data_ptr(), B.data_ptr(), out.data_ptr(), _p(R), M, N, K, A.stride(0), B.stride(0),
                              out.stride(0), ldr, epilogue, tile_cfg, _gemm_ws(A.device) if SPLITK_TAIL else 0, _st())
    _lib.check(rc, "nv_gemm_bf16_ws")
    return out


def gemm_qkv_rope(x, Wqkv, cos_t, sin_t, S, rope_cols, out=None, pos_i32=None):
    """packed q|k|v = x @ Wqkv^T with RoPE on the first `rope_cols` columns applied in the GEMM epilogue; row m sits at position
    pos_i32[m] (packed rows) or m % S"""
    _chk2d(x, BF16)
    _chk2d(Wqkv, BF16)
    M, K = x.shape
    N = Wqkv.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    rc = _L().nv_gemm_bf16_rope(x.data_ptr(), Wqkv.data_ptr(), out.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), _p(pos_i32), M, N, K,
                                x.stride(0), Wqkv.stride(0), out.stride(0), S, rope_cols, _gemm_ws(x.device) if SPLITK_TAIL else 0, _st())
    _lib.check(rc, "nv_gemm_bf16_rope")
    return out


SPLITK_TAIL = True
GEMV_DECODE = True      # M <= 16 NT GEMMs go to nv_gemv_bf16
_gemm_ws_cache = {}


def _gemm_ws(device):
    """zero-initialised split-K workspace, one per (device, stream)"""
    key = _st()
    t = _gemm_ws_cache.get(key)
    if t is None:
        t = torch.zeros((_L().nv_gemm_bf16_workspace_bytes(),), dtype=torch.uint8, device=device)
        _gemm_ws_cache[key] = t
    return t.data_ptr()


# ------------------------------------------------------------------ LM row ops
def embed_vis(table, ids_i32, vis_idx_i32, vis_f32, out=None):
    M, d = ids_i32.numel(), table.shape[1]
    if out is None:
        out = torch.empty((M, d), dtype=BF16, device=table.device)
    rc = _L().nv_embed_vis_bf16(table.data_ptr(), ids_i32.data_ptr(), vis_idx_i32.data_ptr(), _p(vis_f32), out.data_ptr(), M, d,
                                _st())
    _lib.check(rc, "nv_embed_vis_bf16")
    return out


def vis_grad(dE, vis_rows_i32):
    n, d = vis_rows_i32.numel(), dE.shape[1]
    out = torch.empty((n, d), dtype=F32, device=dE.device)
    _lib.check(_L().nv_vis_grad_f32(dE.data_ptr(), vis_rows_i32.data_ptr(), out.data_ptr(), n, d, _st()), "nv_vis_grad_f32")
    return out


def embed_grad(dE, uniq_i32, seg_off_i32, tok_i32, gtable):
    _lib.check(_L().nv_embed_grad_bf16(dE.data_ptr(), uniq_i32.data_ptr(), seg_off_i32.data_ptr(), tok_i32.data_ptr(),
                                       gtable.data_ptr(), uniq_i32.numel(), dE.shape[1], _st()), "nv_embed_grad_bf16")


def rmsnorm_fwd(x, w, eps, out=None, rstd=None):
    M, d = x.shape
    if out is None:
        out = torch.empty_like(x)
    if rstd is None:
        rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.check(_L().nv_rmsnorm_fwd_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, d, eps, _st()),
               "nv_rmsnorm_fwd_bf16")
    return out, rstd


_ws_cache = {}


def _workspace(nbytes, device, tag):
    key = (tag, str(device))
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


def rmsnorm_bwd(dy, x, w, rstd, gw, resid_grad=None, out=None):
    """dx (+ resid_grad) ; weight gradient accumulated into gw (bf16)."""
    M, d = x.shape
    if out is None:
        out = torch.empty_like(x)
    ws = _workspace(_L().nv_rmsnorm_bwd_workspace_bytes(d), x.device, "rms")
    rc = _L().nv_rmsnorm_bwd_bf16(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(resid_grad), out.data_ptr(),
                                  gw.data_ptr(), ws.data_ptr(), M, d, _st())
    _lib.check(rc, "nv_rmsnorm_bwd_bf16")
    return out


def rope_(qkv, cos_t, sin_t, S, H, hd, backward=False):
    M = qkv.shape[0]
    _lib.check(_L().nv_rope_bf16(qkv.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), M, S, H, hd, qkv.stride(0),
                                 1 if backward else 0, _st()), "nv_rope_bf16")
    return qkv


def rope_rows_(qkv, cos_t, sin_t, pos_i32, H, hd):
    """in-place RoPE with an explicit position per row (KV-cache inference)."""
    _lib.check(_L().nv_rope_rows_bf16(qkv.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), pos_i32.data_ptr(), qkv.shape[0], H, hd,
                                      qkv.stride(0), _st()), "nv_rope_rows_bf16")
    return qkv


def swiglu_fwd(gu, out=None):
    M, ff2 = gu.shape
    if out is None:
        out = torch.empty((M, ff2 // 2), dtype=BF16, device=gu.device)
    _lib.check(_L().nv_swiglu_fwd_bf16(gu.data_ptr(), out.data_ptr(), M, ff2 // 2, _st()), "nv_swiglu_fwd_bf16")
    return out


def swiglu_bwd(gu, dh, out=None):
    M, ff2 = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    _lib.check(_L().nv_swiglu_bwd_bf16(gu.data_ptr(), dh.data_ptr(), out.data_ptr(), M, ff2 // 2, _st()), "nv_swiglu_bwd_bf16")
    return out


def scale_bf16_(x, scale):
    _lib.check(_L().nv_scale_bf16(x.data_ptr(), x.data_ptr(), x.numel(), float(scale), _st()), "nv_scale_bf16")
    return x


def scale_dev_bf16(x, scale_dev_f32, out=None, accumulate=False):
    """out = bf16(x * s) or, accumulate: out = bf16(out + bf16(x * s)); s is a 1-element fp32 DEVICE tensor"""
    if out is None:
        assert not accumulate
        out = torch.empty_like(x)
    _lib.check(_L().nv_scale_dev_bf16(x.data_ptr(), out.data_ptr(), x.numel(), scale_dev_f32.data_ptr(), 1 if accumulate else 0, _st()),
               "nv_scale_dev_bf16")
    return out


def gather_rows_bf16(src, rows_i32, out=None):
    n, d = rows_i32.numel(), src.shape[1]
    if out is None:
        out = torch.empty((n, d), dtype=BF16, device=src.device)
    _lib.check(_L().nv_gather_rows_bf16(src.data_ptr(), rows_i32.data_ptr(), out.data_ptr(), n, d, _st()), "nv_gather_rows_bf16")
    return out


def scatter_rows_bf16_(src, rows_i32, dst):
    _lib.check(_L().nv_scatter_rows_bf16(src.data_ptr(), rows_i32.data_ptr(), dst.data_ptr(), rows_i32.numel(), src.shape[1],
                                         _st()), "nv_scatter_rows_bf16")
    return dst


# ------------------------------------------------------------------ attention
def attn_fwd(qkv, kv_start_i32, B, S, H, hd, out=None, lse2=None, q_row_min=0):
    if out is None:
        out = torch.empty((B * S, H * hd), dtype=BF16, device=qkv.device)
    if lse2 is None:
        lse2 = torch.empty((B, H, S), dtype=F32, device=qkv.device)
    rc = _L().nv_attn_fwd_bf16(qkv.data_ptr(), out.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(), B, S, H, hd, q_row_min,
                               _st())
    _lib.check(rc, "nv_attn_fwd_bf16")
    return out, lse2


def attn_fwd_hfround(qkv, kv_start_i32, cu_i32, B, S, H, hd, out, lse2, q_row_min=0):
    """PARITY INSTRUMENT (tests): attention forward with HF eager attention's bf16 rounding points (include/navillm_hip.h)"""
    rc = _L().nv_attn_fwd_hfround_bf16(qkv.data_ptr(), out.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(), _p(cu_i32), B, S, H, hd,
                                       q_row_min, _st())
    _lib.check(rc, "nv_attn_fwd_hfround_bf16")
    return out, lse2


def attn_fwd_strided(qkv, kv_start_i32, B, S, S_stride, H, hd, out, lse2, q_row_min=0):
    """forward over a KV-cache layout (sample b at rows b*S_stride ..), valid length <= S"""
    rc = _L().nv_attn_fwd_strided_bf16(qkv.data_ptr(), out.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(), B, S, S_stride, H,
                                       hd, q_row_min, _st())
    _lib.check(rc, "nv_attn_fwd_strided_bf16")
    return out


def attn_fwd_varlen(qkv, cu_i32, pos0_i32, B, S_max, H, hd, out, lse2, q_row_min=0):
    """packed rows: sample b = rows [cu[b], cu[b+1]); q_row_min = -1: each sample's last 128-row block only"""
    rc = _L().nv_attn_fwd_varlen_bf16(qkv.data_ptr(), out.data_ptr(), lse2.data_ptr(), cu_i32.data_ptr(), pos0_i32.data_ptr(), B, S_max, H,
                                      hd, q_row_min, _st())
    _lib.check(rc, "nv_attn_fwd_varlen_bf16")
    return out, lse2


def attn_bwd_varlen(qkv, out, dout, lse2, cu_i32, pos0_i32, B, S_max, H, hd, dqkv, q_row_min=0, rope=None):
    ws = _workspace(_L().nv_attn_bwd_workspace_bytes(B, S_max, H), qkv.device, "attn")
    rc = _L().nv_attn_bwd_varlen_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), cu_i32.data_ptr(),
                                      pos0_i32.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), _p(rope[0] if rope else None),
                                      _p(rope[1] if rope else None), B, S_max, qkv.shape[0], H, hd, q_row_min, _st())
    _lib.check(rc, "nv_attn_bwd_varlen_bf16")
    return dqkv


def attn_bwd_strided(qkv, out, dout, lse2, kv_start_i32, B, S, S_stride, H, hd, dqkv, q_row_min=0, kv_acc=None, prefix_len_i32=None, first=False):
    """backward over the K/V-cache layout (see nv_attn_bwd_strided_bf16); gradients stay in the rotated frame.
    kv_acc / prefix_len_i32: the K/V gradients of each sample's cached prefix rows go into the fp32 accumulator (first: stored)"""
    ws = _workspace(_L().nv_attn_bwd_workspace_bytes(B, S_stride, H), qkv.device, "attn_strided")
    if kv_acc is not None:
        rc = _L().nv_attn_bwd_strided_kvacc_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(),
                                                 dqkv.data_ptr(), ws.data_ptr(), kv_acc.data_ptr(), prefix_len_i32.data_ptr(), 1 if first else 0,
                                                 B, S, S_stride, H, hd, q_row_min, _st())
        _lib.check(rc, "nv_attn_bwd_strided_kvacc_bf16")
        return dqkv
    rc = _L().nv_attn_bwd_strided_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(),
                                       dqkv.data_ptr(), ws.data_ptr(), B, S, S_stride, H, hd, q_row_min, _st())
    _lib.check(rc, "nv_attn_bwd_strided_bf16")
    return dqkv


def attn_fwd_episode(qkv, out, lse_ptrs_i64, cu_i32, tab_i32, T, B, H, hd, cap, N_max):
    """forward attention of all T steps of a prefix-reuse episode for one layer, in one launch over the episode row buffers
    (nv_attn_fwd_episode_bf16): the steps' rows of `out` and each step's lse (lse_ptrs[t] -> [B, H, cap]) are written"""
    R = qkv.shape[0]
    assert out.shape[0] == R and lse_ptrs_i64.numel() == T and tab_i32.numel() == 2 * T * B and qkv.shape[1] == 3 * H * hd
    assert qkv.is_contiguous() and out.is_contiguous()
    _lib.check(_L().nv_attn_fwd_episode_bf16(qkv.data_ptr(), out.data_ptr(), lse_ptrs_i64.data_ptr(), cu_i32.data_ptr(), tab_i32.data_ptr(),
                                             T, B, H, hd, cap, N_max, R, _st()), "nv_attn_fwd_episode_bf16")
    return out


def attn_bwd_episode(qkv, out, dout, dqkv, lse_ptrs_i64, cu_i32, tab_i32, kv_acc, T, B, H, hd, cap, Mp, Lp_max, N_max, rope=None, accumulate=False):
    """attention backward of all T steps of a prefix-reuse episode for one layer (nv_attn_bwd_episode_acc_bf16): rows [Mp, R) of dqkv and
    the fp32 prefix K/V gradient sums in kv_acc are written (accumulate: added to what an earlier segment of the episode left there);
    rope = (cos, sin) applies RoPE^T to dQ / dK as they are stored"""
    R = qkv.shape[0]
    assert out.shape[0] == R and dout.shape[0] == R and dqkv.shape[0] == R and lse_ptrs_i64.numel() == T and tab_i32.numel() == 2 * T * B
    ws = _workspace(max(R - Mp, 1) * H * 4, qkv.device, "attn_episode")
    cos, sin = (rope[0].data_ptr(), rope[1].data_ptr()) if rope is not None else (0, 0)
    if rope is not None:
        assert rope[0].shape[0] >= cap, "RoPE tables shorter than the K/V cache"
    _lib.check(_L().nv_attn_bwd_episode_acc_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), ws.data_ptr(),
                                                 lse_ptrs_i64.data_ptr(), cu_i32.data_ptr(), tab_i32.data_ptr(), kv_acc.data_ptr(), cos, sin,
                                                 T, B, H, hd, cap, Mp, R, Lp_max, N_max, int(bool(accumulate)), _st()), "nv_attn_bwd_episode_acc_bf16")
    return dqkv


def rope_rows_t_(qkv, cos_t, sin_t, pos_i32, H, hd):
    """in-place RoPE^T with an explicit position per row (backward of rope_rows_)"""
    _lib.check(_L().nv_rope_rows_t_bf16(qkv.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), pos_i32.data_ptr(), qkv.shape[0], H, hd,
                                        qkv.stride(0), _st()), "nv_rope_rows_t_bf16")
    return qkv


def kv_grad_accum(dqkv_full, acc_f32, rows_i32, first=False):
    """first: acc[rows] = ... instead of += (the accumulator then needs no zero-fill)"""
    d = dqkv_full.shape[1] // 3
    if first:
        _lib.check(_L().nv_kv_grad_set_f32(dqkv_full.data_ptr(), acc_f32.data_ptr(), rows_i32.data_ptr(), rows_i32.numel(), d, _st()),
                   "nv_kv_grad_set_f32")
        return
    _lib.check(_L().nv_kv_grad_accum_f32(dqkv_full.data_ptr(), acc_f32.data_ptr(), rows_i32.data_ptr(), rows_i32.numel(), d, _st()),
               "nv_kv_grad_accum_f32")


def kv_grad_inject(dqkv_packed, acc_f32, rows_i32):
    d = dqkv_packed.shape[1] // 3
    _lib.check(_L().nv_kv_grad_inject_bf16(dqkv_packed.data_ptr(), acc_f32.data_ptr(), rows_i32.data_ptr(), rows_i32.numel(), d, _st()),
               "nv_kv_grad_inject_bf16")


def attn_bwd(qkv, out, dout, lse2, kv_start_i32, B, S, H, hd, dqkv=None, q_row_min=0, rope=None):
    """rope=(cos_t, sin_t): dQ/dK leave the kernel already rotated back (RoPE^T fused into the final store)."""
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    ws = _workspace(_L().nv_attn_bwd_workspace_bytes(B, S, H), qkv.device, "attn")
    if rope is None:
        rc = _L().nv_attn_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(),
                                   dqkv.data_ptr(), ws.data_ptr(), B, S, H, hd, q_row_min, _st())
    else:
        rc = _L().nv_attn_bwd_rope_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(), kv_start_i32.data_ptr(),
                                        dqkv.data_ptr(), ws.data_ptr(), rope[0].data_ptr(), rope[1].data_ptr(), B, S, H, hd,
                                        q_row_min, _st())
    _lib.check(rc, "nv_attn_bwd_bf16")
    return dqkv


# ------------------------------------------------------------------ heads / losses / optimizer
def head_fwd(x, W, bias):
    B, d = x.shape
    N = W.shape[0]
    y = torch.empty((B, N), dtype=BF16, device=x.device)
    _lib.check(_L().nv_head_fwd_bf16(x.data_ptr(), W.data_ptr(), bias.data_ptr(), y.data_ptr(), B, d, N, _st()), "nv_head_fwd_bf16")
    return y


def head_bwd(dy, x, W, gW, gb):
    B, d = x.shape
    N = W.shape[0]
    dx = torch.empty_like(x)
    _lib.check(_L().nv_head_bwd_bf16(dy.data_ptr(), x.data_ptr(), W.data_ptr(), dx.data_ptr(), gW.data_ptr(), gb.data_ptr(), B, d,
                                     N, _st()), "nv_head_bwd_bf16")
    return dx


def action_ce(logits, targets_i64, gscale=1.0, want_grad=True, gscale_dev=None):
    """gscale_dev: optional 1-element fp32 DEVICE tensor multiplied into gscale (keeps the coefficient off the host)."""
    B, G = logits.shape
    logits = logits.contiguous()
    loss_rows = torch.empty((B,), dtype=F32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    _lib.check(_L().nv_action_ce_bf16(logits.data_ptr(), targets_i64.data_ptr(), loss_rows.data_ptr(), _p(dl), B, G, gscale,
                                      _p(gscale_dev), _st()), "nv_action_ce_bf16")
    return loss_rows, dl


def lm_ce_(logits, labels_i32, V, special0, nspecial, gscale, write_grad=True, loss_rows=None):
    M = logits.shape[0]
    if loss_rows is None:
        loss_rows = torch.empty((M,), dtype=F32, device=logits.device)
    _lib.check(_L().nv_lm_ce_bf16(logits.data_ptr(), labels_i32.data_ptr(), loss_rows.data_ptr(), M, V, logits.stride(0), special0,
                                  nspecial, gscale, 1 if write_grad else 0, _st()), "nv_lm_ce_bf16")
    return loss_rows


def clip_coef(flat_grads, max_norm, out2=None):
    """flat_grads: list of 1-D bf16/fp32 tensors. Returns device tensor [total_norm, coef]."""
    dev = flat_grads[0].device
    part = _workspace(4 * 2048 * 8 * len(flat_grads), dev, "sumsq").view(torch.float32)
    off = 0
    for g in flat_grads:
        n = ctypes.c_int(0)
        _lib.check(_L().nv_sumsq(g.data_ptr(), g.numel(), 1 if g.dtype == BF16 else 0, part[off:].data_ptr(), ctypes.byref(n),
                                 _st()), "nv_sumsq")
        off += n.value
    if out2 is None:
        out2 = torch.empty((2,), dtype=F32, device=dev)
    _lib.check(_L().nv_clip_coef(part.data_ptr(), off, float(max_norm), out2.data_ptr(), _st()), "nv_clip_coef")
    return out2


def adamw_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01, clip=None, zero_grad=False):
    _lib.check((_L().nv_adamw_zero_grad if zero_grad else _L().nv_adamw)(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1 if p.dtype == BF16 else 0, lr,
                             beta1, beta2, eps, wd, step, _p(clip), _st()), "nv_adamw")


# ------------------------------------------------------------------ fp32 encoder ops
def gemm_f32(layout, A, B, bias=None, out=None, accumulate=False):
    _chk2d(A, F32)
    _chk2d(B, F32)
    if layout == NT:
        M, K = A.shape
        N, K2 = B.shape
    elif layout == NN:
        M, K = A.shape
        K2, N = B.shape
    else:
        K, M = A.shape
        K2, N = B.shape
    assert K == K2
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=F32, device=A.device)
    ws = _workspace(_L().nv_gemm_f32_workspace_bytes(M, N), A.device, "gemm_f32") if M * N <= (1 << 22) else None
    rc = _L().nv_gemm_f32_ws(layout, A.data_ptr(), B.data_ptr(), out.data_ptr(), _p(bias), M, N, K, A.stride(0), B.stride(0),
                             out.stride(0), 1 if accumulate else 0, 0 if ws is None else ws.data_ptr(), _st())
    _lib.check(rc, "nv_gemm_f32_ws")
    return out


def layernorm_fwd(x, w, b, eps):
    M, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.check(_L().nv_layernorm_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         M, d, eps, _st()), "nv_layernorm_fwd_f32")
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, gw=None, gb=None):
    """dx, dgamma, dbeta; when gw/gb are given the parameter gradients are ACCUMULATED into them."""
    M, d = x.shape
    dx = torch.empty_like(x)
    acc = 1 if gw is not None else 0
    if gw is None:
        gw = torch.empty((d,), dtype=F32, device=x.device)
        gb = torch.empty((d,), dtype=F32, device=x.device)
    ws = _workspace(_L().nv_layernorm_bwd_workspace_bytes(d), x.device, "ln")
    _lib.check(_L().nv_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                         gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), M, d, acc, _st()), "nv_layernorm_bwd_f32")
    return dx, gw, gb


def colsum_f32(x, out=None, accumulate=False):
    M, d = x.shape
    if out is None:
        out = torch.empty((d,), dtype=F32, device=x.device)
    _lib.check(_L().nv_colsum_f32(x.data_ptr(), out.data_ptr(), M, d, x.stride(0), 1 if accumulate else 0, _st()), "nv_colsum_f32")
    return out


def _keep_ptr(keep, B, N, heads):
    if keep is None:
        return None
    assert keep.dtype == F32 and keep.is_contiguous() and tuple(keep.shape) == (B, heads, N, N), (keep.shape, keep.dtype)
    return keep.data_ptr()


def mha_fwd(qkv, lens_i32, B, N, heads, hd, p=0.0, seed=0, offset=0, keep=None):
    """-> (out [B*N, h], P [B,heads,N,N] = the probabilities before dropout).  p > 0: attention-probability dropout, mask from
    `keep` (0/1 flags) or from Philox(seed, offset + element)."""
    h = heads * hd
    out = torch.empty((B * N, h), dtype=F32, device=qkv.device)
    P = torch.empty((B, heads, N, N), dtype=F32, device=qkv.device)
    _lib.check(_L().nv_mha_fwd_drop_f32(qkv.data_ptr(), lens_i32.data_ptr(), out.data_ptr(), P.data_ptr(), _keep_ptr(keep, B, N, heads),
                                        float(p), int(seed) & (2 ** 64 - 1), int(offset), B, N, heads, hd, _st()), "nv_mha_fwd_drop_f32")
    return out, P


def mha_bwd(qkv, P, dout, B, N, heads, hd, p=0.0, seed=0, offset=0, keep=None):
    dqkv = torch.empty_like(qkv)
    _lib.check(_L().nv_mha_bwd_drop_f32(qkv.data_ptr(), P.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), _keep_ptr(keep, B, N, heads),
                                        float(p), int(seed) & (2 ** 64 - 1), int(offset), B, N, heads, hd, _st()), "nv_mha_bwd_drop_f32")
    return dqkv


def gelu_fwd(x):
    y = torch.empty_like(x)
    _lib.check(_L().nv_gelu_fwd_f32(x.data_ptr(), y.data_ptr(), x.numel(), _st()), "nv_gelu_fwd_f32")
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    _lib.check(_L().nv_gelu_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _st()), "nv_gelu_bwd_f32")
    return dx


def add_f32(a, b, bcast_rows=False):
    out = torch.empty_like(a)
    d = a.shape[-1]
    _lib.check(_L().nv_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), d, 1 if bcast_rows else 0, _st()), "nv_add_f32")
    return out


def mul_f32(a, b):
    out = torch.empty_like(a)
    _lib.check(_L().nv_mul_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _st()), "nv_mul_f32")
    return out


def dropout_f32(x, p, seed, offset):
    """keep-mask drawn in the kernel from (seed, offset); same call on the gradient = the backward"""
    out = torch.empty_like(x)
    _lib.check(_L().nv_dropout_f32(x.data_ptr(), out.data_ptr(), x.numel(), float(p), int(seed) & (2 ** 64 - 1), int(offset), _st()),
               "nv_dropout_f32")
    return out


def rowscale_f32(x, s):
    """x [rows, d] * s [rows] (fp32 0/1 masks)."""
    out = torch.empty_like(x)
    d = x.shape[-1]
    _lib.check(_L().nv_rowscale_f32(x.data_ptr(), s.data_ptr(), out.data_ptr(), x.numel() // d, d, _st()), "nv_rowscale_f32")
    return out


def gather_add_f32(src, idx_i32, base=None):
    """out[i] = (src[idx[i]] if idx[i] >= 0 else 0) (+ base[i])."""
    rows, d = idx_i32.numel(), src.shape[-1]
    out = torch.empty((rows, d), dtype=F32, device=src.device)
    _lib.check(_L().nv_gather_add_f32(src.data_ptr(), idx_i32.data_ptr(), _p(base), out.data_ptr(), rows, d, _st()),
               "nv_gather_add_f32")
    return out


def index_sum_f32(src, idx_i32, R, out=None, accumulate=False):
    """dst[r] (+)= sum_{i: idx[i]==r} src[i]  for r < R."""
    n, d = src.shape
    dst = out if out is not None else torch.empty((R, d), dtype=F32, device=src.device)
    _lib.check(_L().nv_index_sum_f32(src.data_ptr(), idx_i32.data_ptr(), dst.data_ptr(), n, R, d, 1 if accumulate else 0, _st()),
               "nv_index_sum_f32")
    return dst


def masked_mean_f32(x, mask_f32):
    B, N, d = x.shape
    out = torch.empty((B, d), dtype=F32, device=x.device)
    _lib.check(_L().nv_masked_mean_f32(x.data_ptr(), mask_f32.data_ptr(), out.data_ptr(), B, N, d, _st()), "nv_masked_mean_f32")
    return out
