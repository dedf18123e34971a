"""torch.autograd glue over the HIP ops: autograd is used only as the tape that connects the
stages (scene encoder -> fusion -> LM -> head -> loss); every forward and backward body is a
sequence of libnavillm_hip.so launches (navillm_amd/ops.py)."""
import os

import numpy as np
import torch
from . import debug
from . import ops

F32, BF16 = torch.float32, torch.bfloat16
# A/B knob: 1 = apply SwiGLU' in the down-proj dgrad GEMM's epilogue (dh never reaches HBM) instead of materialising dh and
# running the separate row kernel.  Measured on MI355X (bench.py, same box, 2x each): fused 38.25/38.43 vs separate
# 38.71/38.82 nav-steps/s -- the epilogue's exp/divide work is serialised behind each tile's K loop with the CU's MFMA
# idle, while the separate HBM-bound kernel hides next to the wgrad GEMM of the other stream.  Off by default.
FUSE_SWIGLU_BWD = os.environ.get("NAVILLM_FUSE_SWIGLU_BWD", "0") == "1"
# RoPE applied in the q|k|v projection's GEMM epilogue (bit-identical to the separate in-place pass; A/B knob)
FUSE_ROPE_FWD = os.environ.get("NAVILLM_FUSE_ROPE_FWD", "1") != "0"


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# =============================================================================== fp32 side
def _acc_target(p):
    """Parameters of NavModel carry a persistent `.grad` view into the flat fp32 grad buffer: accumulate
    there directly (no autograd `+=` pass). Plain tensors (tests) get a returned gradient instead."""
    if isinstance(p, torch.nn.Parameter) and p.grad is not None:
        t = getattr(p, "_nv_touch", None)
        if t is not None:
            t()                     # FlatStore.touch(name): this parameter has a gradient from now on (navillm_amd/optim.py)
        return p.grad
    return None


class LinearF32(torch.autograd.Function):
    """y = x W^T + b on [rows, K] (nn.Linear; image_embedding.py:62-66,102; nav_model.py:59-76)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = _c(x).view(-1, x.shape[-1])
        ctx.save_for_backward(x2)
        ctx.W, ctx.b = W, b
        ctx.xshape = x.shape
        y = ops.gemm_f32(ops.NT, x2, W, bias=b)
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        dy2 = _c(dy).view(-1, W.shape[0])
        dx = ops.gemm_f32(ops.NN, dy2, W).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        gW, gb = _acc_target(W), _acc_target(b)
        if gW is not None:
            ops.gemm_f32(ops.TN, dy2, x2, out=gW, accumulate=True)
            dW = None
        else:
            dW = ops.gemm_f32(ops.TN, dy2, x2)
        if gb is not None:
            ops.colsum_f32(dy2, out=gb, accumulate=True)
            db = None
        else:
            db = ops.colsum_f32(dy2)
        return dx, dW, db


class LayerNormF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = _c(x).view(-1, x.shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.w, ctx.b = w, b
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        gw, gb = _acc_target(w), _acc_target(b)
        if gw is not None and gb is not None:
            dx, _, _ = ops.layernorm_bwd(_c(dy).view(x2.shape), x2, w, mean, rstd, gw=gw, gb=gb)
            return dx.view(ctx.xshape), None, None, None
        dx, dw, db = ops.layernorm_bwd(_c(dy).view(x2.shape), x2, w, mean, rstd)
        return dx.view(ctx.xshape), dw, db, None


class GeluF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, _c(dy))


class AddF32(torch.autograd.Function):
    """a + b (same shape) or a + row-broadcast b."""

    @staticmethod
    def forward(ctx, a, b):
        a = _c(a)
        ctx.bcast = (b.dim() == 1)
        ctx.d = a.shape[-1]
        return ops.add_f32(a, _c(b), bcast_rows=ctx.bcast)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        gb = ops.colsum_f32(g.view(-1, ctx.d)) if ctx.bcast else g
        return g, gb


def add(a, b):
    return AddF32.apply(a, b)


class RowScaleF32(torch.autograd.Function):
    """x[rows, d] * s[rows]  (masked_fill(.., 0) with s in {0,1}; dropout with s = keep/(1-p) per element
    uses MulF32)."""

    @staticmethod
    def forward(ctx, x, s):
        x = _c(x)
        ctx.save_for_backward(s)
        return ops.rowscale_f32(x, s)

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return ops.rowscale_f32(_c(g), s), None


class MulMaskF32(torch.autograd.Function):
    """x * mask (dropout; mask already holds keep/(1-p))."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul_f32(_c(x), mask)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return ops.mul_f32(_c(g), mask), None


class DropoutF32(torch.autograd.Function):
    """nn.Dropout(p) as ONE launch: the keep mask comes from Philox4x32-10 inside the kernel, keyed by (seed, offset); the
    backward regenerates it (nothing is stored)."""

    @staticmethod
    def forward(ctx, x, p, seed, offset):
        ctx.key = (p, seed, offset)
        return ops.dropout_f32(_c(x), p, seed, offset)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout_f32(_c(g), *ctx.key), None, None, None


def _philox_take(device, n):
    """(seed, offset) for n draws from torch's own CUDA generator state, advanced like a torch kernel would (Philox counter
    space in units of 4 draws): `torch.manual_seed` re-keys and rewinds it, exactly as it does for torch's dropout."""
    g = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = g.get_offset()
    g.set_offset(off + 4 * ((n + 3) // 4))
    return g.initial_seed(), off // 4


def dropout(x, p, training, mask=None):
    """nn.Dropout. `mask` (0/1 keep flags) may be injected for parity tests, otherwise it is drawn inside the HIP kernel."""
    if not training or p == 0.0:
        return x
    if mask is not None:
        return MulMaskF32.apply(x, mask.to(torch.float32) * (1.0 / (1.0 - p)))
    seed, off = _philox_take(x.device, x.numel())
    return DropoutF32.apply(x, p, seed, off)


class EmbedAddF32(torch.autograd.Function):
    """base + table[idx]  (idx < 0 -> +0).  nn.Embedding lookups of the encoder/fusion."""

    @staticmethod
    def forward(ctx, table, idx_i32, base):
        ctx.save_for_backward(idx_i32)
        ctx.table = table
        ctx.has_base = base is not None
        b2 = _c(base).view(-1, table.shape[1]) if base is not None else None
        out = ops.gather_add_f32(table, idx_i32, b2)
        return out.view(*base.shape) if base is not None else out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g2 = _c(g).view(idx.numel(), -1)
        gt = _acc_target(ctx.table)
        if gt is not None:
            ops.index_sum_f32(g2, idx, ctx.table.shape[0], out=gt, accumulate=True)
            dt = None
        else:
            dt = ops.index_sum_f32(g2, idx, ctx.table.shape[0])
        return dt, None, (g if ctx.has_base else None)


class GatherRowsF32(torch.autograd.Function):
    """out[i] = src[idx[i]] (idx<0 -> 0) (+ base[i]); src rows may be hit at most once (fusion /
    candidate selection), so the backward is a gather with the inverse table."""

    @staticmethod
    def forward(ctx, src, idx_i32, inv_idx_i32, base):
        src2 = _c(src).view(-1, src.shape[-1])
        ctx.save_for_backward(idx_i32, inv_idx_i32)
        ctx.sshape = src.shape
        ctx.has_base = base is not None
        b2 = _c(base).view(-1, src.shape[-1]) if base is not None else None
        return ops.gather_add_f32(src2, idx_i32, b2)

    @staticmethod
    def backward(ctx, g):
        idx, inv = ctx.saved_tensors
        g = _c(g)
        ds = ops.gather_add_f32(g, inv, None).view(ctx.sshape)
        return ds, None, None, (g if ctx.has_base else None)


class MHAF32(torch.autograd.Function):
    """dropout_p(softmax(q k^T / sqrt(hd) + key_padding)) v per head on packed qkv (nn.MultiheadAttention core incl. its
    attention-probability dropout, detr_transformer.py:138).  `drop` = None (eval / p = 0) or (p, seed, offset, keep): keep is an
    injected [B,heads,N,N] 0/1 mask (parity tests) or None = drawn in the kernel from Philox(seed, offset + element) and
    regenerated by the backward."""

    @staticmethod
    def forward(ctx, qkv, lens_i32, B, N, heads, hd, drop=None):
        qkv = _c(qkv)
        p, seed, off, keep = drop if drop is not None else (0.0, 0, 0, None)
        out, P = ops.mha_fwd(qkv, lens_i32, B, N, heads, hd, p, seed, off, keep)
        ctx.save_for_backward(qkv, P)
        ctx.dims = (B, N, heads, hd)
        ctx.drop = (p, seed, off, keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, P = ctx.saved_tensors
        return ops.mha_bwd(qkv, P, _c(dout), *ctx.dims, *ctx.drop), None, None, None, None, None, None


def mha(qkv, lens_i32, B, N, heads, hd, p, training, mask=None):
    """nn.MultiheadAttention's core with its dropout on the attention probabilities (training only)."""
    if not training or p == 0.0:
        return MHAF32.apply(qkv, lens_i32, B, N, heads, hd, None)
    if mask is not None:
        return MHAF32.apply(qkv, lens_i32, B, N, heads, hd, (p, 0, 0, mask.to(torch.float32).contiguous()))
    n = B * heads * N * N
    seed, off = _philox_take(qkv.device, 4 * n)        # one Philox counter per probability element
    return MHAF32.apply(qkv, lens_i32, B, N, heads, hd, (p, seed, off, None))


def linear(x, W, b):
    return LinearF32.apply(x, W, b)


def layer_norm(x, w, b, eps):
    return LayerNormF32.apply(x, w, b, eps)


# =============================================================================== bf16 LM side
class EmbedVis(torch.autograd.Function):
    """E = embed_tokens[ids]; E[special] = bf16(f32(E) + vis)   (modified_lm.py:100-110)."""

    @staticmethod
    def forward(ctx, vis, anchor, model, ids_i32, vis_idx_i32, vis_rows_i32, ids_cpu):
        st = model.store
        table = st.p("lang_model.model.embed_tokens.weight")
        # into the arena: with packed rows the row count changes from step to step, and a 40 MB torch allocation of a new
        # size now and then costs a hipMalloc stall (seen as 95-110 ms forward steps among 64 ms ones)
        M = ids_i32.numel()
        model.arena.reserve(M)
        E = ops.embed_vis(table, ids_i32, vis_idx_i32, vis, out=model.arena.scratch["E"][:M])
        ctx.model = model
        ctx.vis_rows = vis_rows_i32
        ctx.has_vis = vis is not None
        ctx.group = None
        if anchor is not None:
            # table gradient: token rows grouped by id.  The ids are host data (tokenizer output) and known NOW, so the
            # grouping (numpy) and its upload happen in the forward, off the backward's critical path
            ids = np.asarray(ids_cpu).reshape(-1)
            order = np.argsort(ids, kind="stable")
            uniq, counts = np.unique(ids, return_counts=True)
            seg = np.zeros(uniq.size + 1, dtype=np.int32)
            np.cumsum(counts, out=seg[1:])
            dev = E.device
            ctx.group = (ops.h2d(torch.from_numpy(uniq.astype(np.int32)), dev), ops.h2d(torch.from_numpy(seg), dev),
                         ops.h2d(torch.from_numpy(order.astype(np.int32)), dev))
        return E

    @staticmethod
    def backward(ctx, dE):
        dE = _c(dE)
        st = ctx.model.store
        dvis = ops.vis_grad(dE, ctx.vis_rows) if ctx.has_vis else None
        uniq, seg, order = ctx.group
        ops.embed_grad(dE, uniq, seg, order, st.g("lang_model.model.embed_tokens.weight"))
        st.touch("lang_model.model.embed_tokens.weight")
        return dvis, None, None, None, None, None, None


class ActivationArena:
    """Persistent HBM arena for the decoder stack's saved activations and backward scratch.

    MI355X-first memory plan: with 288 GB per GPU there is no reason to churn a caching allocator with
    ~350 differently-sized tensors per step whose sizes change whenever the prompt grows by one history
    token (that churn cost 25% of the step in hipMalloc stalls, profiles/r01_*).  One slab, sized for
    M_cap = B*S rounded up to 2048 rows, carved once into per-layer views; a step just slices [:M].
    A later LM forward reuses the slab, so backward must run before the next forward of the same model
    (true for the rollout loop: every loss is followed by backward(), mp3d_agent.py:756,824,866,902)."""

    LAYER_FIELDS = (("n1", 1, 0), ("qkv", 3, 0), ("attn", 1, 0), ("x1", 1, 0), ("n2", 1, 0), ("gu", 0, 2), ("h", 0, 1),
                    ("x2", 1, 0))
    SCRATCH_FIELDS = (("dh", 0, 1), ("dgu", 0, 2), ("dn2", 1, 0), ("dx1", 1, 0), ("dattn", 1, 0), ("dqkv", 3, 0),
                      ("dn1", 1, 0), ("dxa", 1, 0), ("dxb", 1, 0), ("E", 1, 0), ("dE", 1, 0))

    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self.cap = 0
        self.generation = 0
        self.layers, self.scratch = [], {}

    def reserve(self, M):
        if M <= self.cap:
            return
        cfg = self.cfg
        cap = (M + 2047) // 2048 * 2048
        d, ff, L, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads
        self.layers, self.scratch, self.slab = [], {}, None      # drop the old slab before allocating the new one
        per_layer = sum(cap * (a * d + b * ff) for _, a, b in self.LAYER_FIELDS)
        per_scr = sum(cap * (a * d + b * ff) for _, a, b in self.SCRATCH_FIELDS)
        self.slab = torch.empty(L * per_layer + per_scr, dtype=BF16, device=self.device)
        self.fslab = torch.empty(L * (2 * cap + cap * H), dtype=F32, device=self.device)
        off = 0
        foff = 0
        for _ in range(L):
            rec = {}
            for name, a, b in self.LAYER_FIELDS:
                w = a * d + b * ff
                rec[name] = self.slab[off:off + cap * w].view(cap, w)
                off += cap * w
            rec["rstd1"] = self.fslab[foff:foff + cap]; foff += cap
            rec["rstd2"] = self.fslab[foff:foff + cap]; foff += cap
            rec["lse"] = self.fslab[foff:foff + cap * H]; foff += cap * H
            self.layers.append(rec)
        for name, a, b in self.SCRATCH_FIELDS:
            w = a * d + b * ff
            self.scratch[name] = self.slab[off:off + cap * w].view(cap, w)
            off += cap * w
        self.cap = cap

    def poison(self, keep=("E",)):
        """NAVILLM_POISON=1: the next forward takes the arena over -- everything the previous one left (saved activations, backward scratch)
        becomes NaN, except the embedding rows `EmbedVis.forward` has just written for THIS forward"""
        if not debug.POISON or not self.cap:
            return
        for rec in self.layers:
            for t in rec.values():
                debug.poison_(t)
        for k, t in self.scratch.items():
            if k not in keep:
                debug.poison_(t)


class LlamaStack(torch.autograd.Function):
    """All decoder layers + final RMSNorm (HF LlamaModel reached from modified_lm.py:112-116).
    Weight gradients are accumulated straight into the flat grad buffer by GEMM epilogues; activations
    live in the model's ActivationArena."""

    @staticmethod
    def forward(ctx, E, model, B, S, kv_start_i32, tail_rows=None, packed=None):
        """tail_rows (int32 device tensor, one row per sample, each the LAST position of its sample) switches on
        the pruned last layer: only those rows are consumed downstream (nav_model.py:237 reads the <cls_1> row), so
        after the last layer's K/V projection everything -- attention queries, o_proj, MLP, final norm -- is
        computed for B rows instead of B*S.  Returns [B, d] then.  (Same spirit as skipping lm_head in navigation
        mode: work the reference does and never reads.)

        packed = (cu_i32 [B+1], pos_i32 [M], S_max): E holds only the REAL tokens of the batch, sample b in rows
        [cu[b], cu[b+1]) (the left-padding rows of the reference's [B, S] layout are never materialised: they feed nothing,
        every row kernel and GEMM here is row-wise, and attention masks them anyway); pos = each row's position in the
        reference's frame (it numbers positions over the padding), kv_start = position of each sample's first token."""
        cfg, st, ar = model.cfg, model.store, model.arena
        H, hd, eps = cfg.num_heads, cfg.head_dim, cfg.rms_norm_eps
        if model.fp8 is not None and ctx.needs_input_grad[0]:
            raise RuntimeError("the weight-only fp8 model is inference only: run it under torch.no_grad()")
        M = E.shape[0]
        cu, pos, Sm = packed if packed is not None else (None, None, S)
        ar.reserve(max(M, B * Sm))
        ar.generation += 1
        if debug.POISON and E.data_ptr() == ar.scratch["E"].data_ptr():
            ar.poison()
        x = E

        def qkv_proj(n1, i, a):
            if FUSE_ROPE_FWD and M > 16:
                return ops.gemm_qkv_rope(n1, model.lm_w(i, "qkv"), model.rope_cos, model.rope_sin, S, 2 * H * hd, out=a["qkv"][:M], pos_i32=pos)
            qkv = model.lm_linear(n1, i, "qkv", out=a["qkv"][:M])
            if pos is not None:
                return ops.rope_rows_(qkv, model.rope_cos, model.rope_sin, pos, H, hd)
            return ops.rope_(qkv, model.rope_cos, model.rope_sin, S, H, hd)

        def attention(qkv, a, qmin):
            lse = a["lse"][:B * H * Sm].view(B, H, Sm)
            if model.attn_hf_rounding:       # parity instrument (tests): HF eager attention's rounding points, forward only
                if cu is not None:
                    return ops.attn_fwd_hfround(qkv, kv_start_i32, cu, B, Sm, H, hd, a["attn"][:M], lse, q_row_min=qmin)
                return ops.attn_fwd_hfround(qkv, kv_start_i32, None, B, S, H, hd, a["attn"][:M], lse, q_row_min=max(qmin, 0))
            if cu is not None:
                return ops.attn_fwd_varlen(qkv, cu, kv_start_i32, B, Sm, H, hd, out=a["attn"][:M], lse2=lse, q_row_min=qmin)
            return ops.attn_fwd(qkv, kv_start_i32, B, S, H, hd, out=a["attn"][:M], lse2=lse, q_row_min=max(qmin, 0))

        L_full = cfg.num_layers - (1 if tail_rows is not None else 0)
        for i in range(L_full):
            p = f"lang_model.model.layers.{i}."
            a = ar.layers[i]
            n1, rstd1 = ops.rmsnorm_fwd(x, st.p(p + "input_layernorm.weight"), eps, out=a["n1"][:M], rstd=a["rstd1"][:M])
            qkv = qkv_proj(n1, i, a)
            attn, lse = attention(qkv, a, 0)
            x1 = model.lm_linear(attn, i, "o", out=a["x1"][:M], R=x, epilogue=ops.EPI_RESID)
            n2, rstd2 = ops.rmsnorm_fwd(x1, st.p(p + "post_attention_layernorm.weight"), eps, out=a["n2"][:M],
                                        rstd=a["rstd2"][:M])
            gu = model.lm_linear(n2, i, "gate_up", out=a["gu"][:M])
            h = ops.swiglu_fwd(gu, out=a["h"][:M])
            x = model.lm_linear(h, i, "down", out=a["x2"][:M], R=x1, epilogue=ops.EPI_RESID)
        ctx.tail = None
        if tail_rows is not None:
            i = cfg.num_layers - 1
            p = f"lang_model.model.layers.{i}."
            a = ar.layers[i]
            # padded layout: every sample ends at row S-1 -> one global first query row; packed: per sample (-1)
            qmin = -1 if cu is not None else ((S - 1) // 128) * 128
            n1, _ = ops.rmsnorm_fwd(x, st.p(p + "input_layernorm.weight"), eps, out=a["n1"][:M], rstd=a["rstd1"][:M])
            qkv = qkv_proj(n1, i, a)
            attn, _ = attention(qkv, a, qmin)
            attn_r = ops.gather_rows_bf16(attn, tail_rows)
            x_r = ops.gather_rows_bf16(x, tail_rows)
            x1_r = model.lm_linear(attn_r, i, "o", R=x_r, epilogue=ops.EPI_RESID)
            n2_r, rstd2_r = ops.rmsnorm_fwd(x1_r, st.p(p + "post_attention_layernorm.weight"), eps)
            gu_r = model.lm_linear(n2_r, i, "gate_up")
            h_r = ops.swiglu_fwd(gu_r)
            x = model.lm_linear(h_r, i, "down", R=x1_r, epilogue=ops.EPI_RESID)
            ctx.tail = (tail_rows, qmin, attn_r, x1_r, rstd2_r, n2_r, gu_r, h_r, x)
        Hs, rstdf = ops.rmsnorm_fwd(x, st.p("lang_model.model.norm.weight"), eps)
        ctx.model, ctx.E, ctx.rstdf = model, E, rstdf
        ctx.dims = (B, S, kv_start_i32, packed, M)
        ctx.generation = ar.generation
        return Hs

    @staticmethod
    def backward(ctx, dH):
        model = ctx.model
        cfg, st, ar = model.cfg, model.store, model.arena
        if ctx.generation != ar.generation:
            raise RuntimeError("the LM activation arena was reused by a later forward before this backward ran; "
                               "call backward() right after each loss (as the rollout loop does)")
        B, S, kvs, packed, M = ctx.dims
        cu, pos, Sm = packed if packed is not None else (None, None, S)
        rope = (model.rope_cos, model.rope_sin)

        def attention_bwd(qkv, attn, dattn, lse, dqkv, qmin):
            if cu is not None:
                return ops.attn_bwd_varlen(qkv, attn, dattn, lse, cu, kvs, B, Sm, H, hd, dqkv, q_row_min=qmin, rope=rope)
            return ops.attn_bwd(qkv, attn, dattn, lse, kvs, B, S, H, hd, dqkv=dqkv, q_row_min=max(qmin, 0), rope=rope)

        H, hd, L = cfg.num_heads, cfg.head_dim, cfg.num_layers
        sc = {k: v[:M] for k, v in ar.scratch.items()}
        # round 5: the first backward after a zero_grad() (FlatStore.layers_zero) STORES its weight gradients -- every matrix is written
        # exactly once per backward -- instead of read-accumulating zeros; every later backward of the optimizer step read-adds
        WACC = ops.EPI_STORE if (getattr(st, "layers_zero", False) and os.environ.get("NAVILLM_WGRAD_STORE", "1") != "0") else ops.EPI_ACCUM
        if WACC == ops.EPI_STORE and debug.POISON:
            st.assert_layers_zero("LlamaStack.backward")
        st.touch_layers()
        model._dp_begin_backward()
        L_full = L
        if ctx.tail is not None:
            # pruned last layer: B-row backward of final norm / MLP / o_proj, then the full-size K/V side
            rows, qmin, attn_r, x1_r, rstd2_r, n2_r, gu_r, h_r, x2_r = ctx.tail
            i = L - 1
            L_full = L - 1
            p = f"lang_model.model.layers.{i}."
            a = ar.layers[i]
            x = ctx.E if i == 0 else ar.layers[i - 1]["x2"][:M]
            n1, qkv, attn = a["n1"][:M], a["qkv"][:M], a["attn"][:M]
            lse = a["lse"][:B * H * Sm].view(B, H, Sm)
            dx2_r = ops.rmsnorm_bwd(_c(dH), x2_r, st.p("lang_model.model.norm.weight"), ctx.rstdf,
                                    st.g("lang_model.model.norm.weight"))
            dh_r = ops.gemm_bf16(ops.NN, dx2_r, st.p(p + "mlp.down_proj.weight"))
            ops.gemm_bf16(ops.TN, dx2_r, h_r, out=st.g(p + "mlp.down_proj.weight"), epilogue=WACC)
            dgu_r = ops.swiglu_bwd(gu_r, dh_r)
            dn2_r = ops.gemm_bf16(ops.NN, dgu_r, st.gate_up(i))
            ops.gemm_bf16(ops.TN, dgu_r, n2_r, out=st.gate_up(i, grad=True), epilogue=WACC)
            dx1_r = ops.rmsnorm_bwd(dn2_r, x1_r, st.p(p + "post_attention_layernorm.weight"), rstd2_r,
                                    st.g(p + "post_attention_layernorm.weight"), resid_grad=dx2_r)
            dattn_r = ops.gemm_bf16(ops.NN, dx1_r, st.p(p + "self_attn.o_proj.weight"))
            ops.gemm_bf16(ops.TN, dx1_r, attn_r, out=st.g(p + "self_attn.o_proj.weight"), epilogue=WACC)
            dattn = sc["dattn"]
            dattn.zero_()
            ops.scatter_rows_bf16_(dattn_r, rows, dattn)
            dqkv = sc["dqkv"]
            dqkv.zero_()                       # dQ rows below qmin are not written by the kernel
            attention_bwd(qkv, attn, dattn, lse, dqkv, qmin)
            dn1 = ops.gemm_bf16(ops.NN, dqkv, st.qkv(i), out=sc["dn1"])
            ops.gemm_bf16(ops.TN, dqkv, n1, out=st.qkv(i, grad=True), epilogue=WACC)
            resid = sc["dx1"]
            resid.zero_()
            ops.scatter_rows_bf16_(dx1_r, rows, resid)
            dx = ops.rmsnorm_bwd(dn1, x, st.p(p + "input_layernorm.weight"), a["rstd1"][:M], st.g(p + "input_layernorm.weight"),
                                 resid_grad=resid, out=sc["dxa"])
            model._dp_layer_done(i, [])
        else:
            xL = ar.layers[L - 1]["x2"][:M]
            dx = ops.rmsnorm_bwd(_c(dH), xL, st.p("lang_model.model.norm.weight"), ctx.rstdf,
                                 st.g("lang_model.model.norm.weight"), out=sc["dxa"])
        nxt = sc["dxb"]

        # Two HIP streams: the dgrad chain (critical path) stays on the current stream; every wgrad GEMM goes to
        # a side stream.  A wgrad has no consumer before the optimizer / the DP all-reduce, so it can trail the
        # chain by a layer: its blocks fill the CUs that a dgrad GEMM's last, partial round leaves idle, and it
        # overlaps the HBM-bound kernels of the chain (RMSNorm/SwiGLU/RoPE backward) that cannot feed the MFMAs.
        # Hazards are on the scratch buffers only (dY tensors are recycled one layer later): `last_read[buf]` is
        # the side-stream event after which `buf` may be overwritten by the chain.
        main = torch.cuda.current_stream()
        side = model.wgrad_stream() if model.overlap_wgrad else None
        last_read = {}

        # NAVILLM_OVERLAP_WGRAD=2: the wgrad of an op runs CONCURRENTLY with its dgrad and the chain joins both before its next
        # kernel (one kernel boundary fewer per pair, the dgrad's partial last round filled).  Measured -1.7 % (43.97 -> 43.22
        # nav-steps/s, ABAB): two 256x256-tile GEMMs sharing the CUs evict each other's panels from the XCD L2s.
        pair = side is not None and model.overlap_wgrad == 2

        def mark():
            """(pair mode) event on the chain before a dgrad GEMM is launched: its dY is ready from here on"""
            if not pair:
                return None
            e = torch.cuda.Event()
            e.record(main)
            return e

        def wgrad(dy_name, dy, act, gout, pre=None):
            if side is None:
                ops.gemm_bf16(ops.TN, dy, act, out=gout, epilogue=WACC)
                return None
            if pair:
                side.wait_event(pre)
                with torch.cuda.stream(side):
                    ops.gemm_bf16(ops.TN, dy, act, out=gout, epilogue=WACC)
                    done = torch.cuda.Event()
                    done.record(side)
                main.wait_event(done)                          # the chain's next kernel starts after BOTH GEMMs
                return None
            e = torch.cuda.Event()
            e.record(main)
            side.wait_event(e)
            with torch.cuda.stream(side):
                ops.gemm_bf16(ops.TN, dy, act, out=gout, epilogue=WACC)
                done = torch.cuda.Event()
                done.record(side)
            last_read[dy_name] = done
            return done

        def before_write(name):
            ev = last_read.pop(name, None)
            if ev is not None:
                main.wait_event(ev)

        dx_name, nxt_name = "dxa", "dxb"
        for i in reversed(range(L_full)):
            p = f"lang_model.model.layers.{i}."
            a = ar.layers[i]
            x = ctx.E if i == 0 else ar.layers[i - 1]["x2"][:M]
            n1, qkv, attn, x1, n2, gu, h = (a[k][:M] for k in ("n1", "qkv", "attn", "x1", "n2", "gu", "h"))
            lse = a["lse"][:B * H * Sm].view(B, H, Sm)
            Wd, Wo = st.p(p + "mlp.down_proj.weight"), st.p(p + "self_attn.o_proj.weight")
            if FUSE_SWIGLU_BWD:
                # dh = dx @ Wd never reaches HBM: the dgrad GEMM's epilogue applies SwiGLU' and writes d(gate|up) directly
                before_write("dgu")
                dgu = ops.gemm_bf16(ops.NN, dx, Wd, out=sc["dgu"], R=gu, epilogue=ops.EPI_SWIGLU_BWD)
                ev = [wgrad(dx_name, dx, h, st.g(p + "mlp.down_proj.weight"))]
            else:
                pre = mark()
                dh = ops.gemm_bf16(ops.NN, dx, Wd, out=sc["dh"])
                ev = [wgrad(dx_name, dx, h, st.g(p + "mlp.down_proj.weight"), pre)]
                before_write("dgu")
                dgu = ops.swiglu_bwd(gu, dh, out=sc["dgu"])
            pre = mark()
            dn2 = ops.gemm_bf16(ops.NN, dgu, st.gate_up(i), out=sc["dn2"])
            ev.append(wgrad("dgu", dgu, n2, st.gate_up(i, grad=True), pre))
            before_write("dx1")
            dx1 = ops.rmsnorm_bwd(dn2, x1, st.p(p + "post_attention_layernorm.weight"), a["rstd2"][:M],
                                  st.g(p + "post_attention_layernorm.weight"), resid_grad=dx, out=sc["dx1"])
            pre = mark()
            dattn = ops.gemm_bf16(ops.NN, dx1, Wo, out=sc["dattn"])
            ev.append(wgrad("dx1", dx1, attn, st.g(p + "self_attn.o_proj.weight"), pre))
            before_write("dqkv")
            dqkv = attention_bwd(qkv, attn, dattn, lse, sc["dqkv"], 0)
            pre = mark()
            dn1 = ops.gemm_bf16(ops.NN, dqkv, st.qkv(i), out=sc["dn1"])
            ev.append(wgrad("dqkv", dqkv, n1, st.qkv(i, grad=True), pre))
            before_write(nxt_name)
            ndx = ops.rmsnorm_bwd(dn1, x, st.p(p + "input_layernorm.weight"), a["rstd1"][:M], st.g(p + "input_layernorm.weight"),
                                  resid_grad=dx1, out=nxt)
            dx, nxt = ndx, dx
            dx_name, nxt_name = nxt_name, dx_name
            model._dp_layer_done(i, [e_ for e_ in ev if e_ is not None])
        out = sc["dE"]
        out.copy_(dx)
        if side is not None:
            main.wait_stream(side)      # weight gradients complete before anything downstream (optimizer, next forward)
        return out, None, None, None, None, None, None


class GatherRowsBF16(torch.autograd.Function):
    """rows of the hidden states (the <cls_1> positions, nav_model.py:237)."""

    @staticmethod
    def forward(ctx, Hs, rows_i32):
        ctx.save_for_backward(rows_i32)
        ctx.shape = Hs.shape
        return ops.gather_rows_bf16(Hs, rows_i32)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        dH = torch.zeros(ctx.shape, dtype=BF16, device=g.device)
        ops.scatter_rows_bf16_(_c(g), rows, dH)
        return dH, None


class HeadBF16(torch.autograd.Function):
    """out_head / og_head Linear(d -> 100) in the LM dtype; weight grads go to the flat buffer."""

    @staticmethod
    def forward(ctx, x, model, prefix):
        st = model.store
        W, b = st.p(prefix + ".weight"), st.p(prefix + ".bias")
        ctx.save_for_backward(x)
        ctx.model, ctx.prefix = model, prefix
        return ops.head_fwd(x, W, b)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        st = ctx.model.store
        pf = ctx.prefix
        dx = ops.head_bwd(_c(dy), x, st.p(pf + ".weight"), st.g(pf + ".weight"), st.g(pf + ".bias"))
        st.touch(pf + ".weight", pf + ".bias")
        return dx, None, None


LM_HEAD_CHUNK_ROWS = int(os.environ.get("NAVILLM_LMHEAD_CHUNK", "2048"))


class LMHeadLoss(torch.autograd.Function):
    """K9: lm_head + special-id mask + shifted mean CE (modified_lm.py:119-137) WITHOUT the [M, V] logits.

    The token rows go through in chunks of LM_HEAD_CHUNK_ROWS: logits of one chunk ([2048, 32064] bf16 = 131 MB, which stays
    in the 256 MB Infinity Cache) -> masked log-softmax + NLL, overwritten in place by d(logits) = (p - onehot) / n_valid ->
    the chunk's dH = dl W and dW += dl^T H right away.  The full logits (M x 32006: 320 MB at B=8, S=650, and three more
    passes over them) never exist.  Because the loss's upstream gradient g (the caller's `* gen_loss_coef / B / accum`,
    mp3d_agent.py:865,901; llava.py:38-40) is not known in the forward, dH and dW are formed for g = 1 -- dW in a scratch
    buffer -- and the backward applies g from DEVICE memory: dH * g, grad_W += g * dW (no host sync, no logits re-read)."""

    @staticmethod
    def forward(ctx, Hs, model, labels_shift_i32, n_valid):
        cfg, st = model.cfg, model.store
        W = st.lm_head_padded()                 # [V_pad, d], pad rows are zero
        V, Vp, d = cfg.vocab_size, W.shape[0], W.shape[1]
        M = Hs.shape[0]
        need = ctx.needs_input_grad[0]
        CH = min(LM_HEAD_CHUNK_ROWS, max(M, 1))
        ws = getattr(model, "_lmhead_ws", None)
        if ws is None or ws["buf"].shape[0] < CH:
            ws = model._lmhead_ws = {"buf": torch.empty((CH, Vp), dtype=BF16, device=Hs.device), "dW": None}
        rows = torch.empty((M,), dtype=F32, device=Hs.device)
        dH = dW = None
        if need:
            if ws["dW"] is None:
                ws["dW"] = torch.empty((Vp, d), dtype=BF16, device=Hs.device)
            dW = ws["dW"]
            dH = torch.empty_like(Hs)
        inv = 1.0 / max(n_valid, 1)
        for c0 in range(0, M, CH):
            c1 = min(M, c0 + CH)
            hc = Hs[c0:c1]
            lg = ops.gemm_bf16(ops.NT, hc, W, out=ws["buf"][:c1 - c0])           # pad columns are exactly 0
            ops.lm_ce_(lg, labels_shift_i32[c0:c1], V, cfg.special_token_ids[0], len(cfg.special_token_ids), inv,
                       write_grad=need, loss_rows=rows[c0:c1])
            if need:
                ops.gemm_bf16(ops.NN, lg, W, out=dH[c0:c1])
                ops.gemm_bf16(ops.TN, lg, hc, out=dW, epilogue=ops.EPI_STORE if c0 == 0 else ops.EPI_ACCUM)
        loss = (rows.sum() / max(n_valid, 1)).to(BF16)
        ctx.model, ctx.dH, ctx.gen = model, dH, None
        if need:
            ws["gen"] = ws.get("gen", 0) + 1
            ctx.gen = ws["gen"]
        return loss

    @staticmethod
    def backward(ctx, g):
        ws, st = ctx.model._lmhead_ws, ctx.model.store
        if ctx.gen != ws.get("gen"):
            raise RuntimeError("the lm_head gradient scratch was reused by a later LM-loss forward before this backward ran; "
                               "call backward() right after each loss (as the rollout loop does)")
        gs = g.reshape(1).float()
        ops.scale_dev_bf16(ws["dW"], gs, out=st.lm_head_padded(grad=True), accumulate=True)
        st.touch("lang_model.lm_head.weight")
        return ops.scale_dev_bf16(ctx.dH, gs, out=ctx.dH), None, None, None


class ActionCE(torch.autograd.Function):
    """CrossEntropyLoss(ignore_index=-100, reduction='sum') on bf16 logits (train.py:229).
    The upstream gradient (the loss coefficient, mp3d_agent.py:750) stays on the device: no host sync."""

    @staticmethod
    def forward(ctx, logits, targets):
        logits = _c(logits)
        rows, _ = ops.action_ce(logits, targets, want_grad=False)
        ctx.save_for_backward(logits, targets)
        return rows.sum().to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        logits, targets = ctx.saved_tensors
        _, dl = ops.action_ce(logits, targets, gscale=1.0, gscale_dev=g.reshape(1).float(), want_grad=True)
        return dl, None
