"""Inference-only incremental LM over a per-layer K/V cache in HBM.

Two callers:

* **prefix reuse across navigation steps** (`NavModel.enable_kv_cache`): the prompt of step t+1 repeats the prompt
  of step t up to the last `<hist>` token (tasks/datasets/r2r.py:18-22; history embeddings are detached constants,
  tasks/agents/mp3d_agent.py:778), so only the new suffix (~100 tokens instead of ~650) goes through the decoder.
* **greedy generation** (`KVCacheLM.generate`): HF `generate(do_sample=False)` as the reference calls it
  (models/nav_model.py:324-341,388-402; models/modified_lm.py:184-199) with the special-id logit mask
  (modified_lm.py:122-124) and the trie-constrained decoding of modified_lm.py:10-30.

Layout: one packed post-RoPE `qkv` buffer per decoder layer, `[B*capacity (+1 junk row), 3d]` bf16 (7B, B=8,
capacity 1024: 201 MB per layer, 6.4 GB in all -- nothing next to 288 GB).  Samples are LEFT-aligned (token j of
sample b lives at row b*capacity + j, position j): appending never moves cached rows, and the positions are the
`position_ids = cumsum(attention_mask)-1` HF's generation path uses.  The training/prefill path of the reference
numbers positions over the left padding instead; RoPE scores depend on position differences only, so the two frames
agree up to bf16 rounding of the rotated q/k (tests/test_kvcache_gpu.py bounds it against full recompute).

Everything arithmetic is a launch of the same HIP kernels as the training path (plus `nv_rope_rows_bf16` and
`nv_attn_fwd_strided_bf16`); there is no autograd here.
"""
import torch

from . import ops

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class KVCacheLM:
    def __init__(self, model, batch_size, capacity=1024):
        cfg = model.cfg
        self.model, self.B, self.cap = model, batch_size, capacity
        d = cfg.hidden_size
        dev = model.device
        rows = batch_size * capacity + 1                      # + one junk row that padding rows of a block scatter into
        self.qkv = [torch.zeros((rows, 3 * d), dtype=BF16, device=dev) for _ in range(cfg.num_layers)]
        self.attn = torch.zeros((batch_size * capacity, d), dtype=BF16, device=dev)
        self.lse = torch.empty((batch_size, cfg.num_heads, capacity), dtype=F32, device=dev)
        self.kv0 = torch.zeros((batch_size,), dtype=I32, device=dev)
        self.reset()

    def reset(self, b=None):
        """forget the cached tokens (of sample b, or of every sample) -- e.g. at an episode boundary"""
        if b is None:
            self.ids = [[] for _ in range(self.B)]
            self.keys = [[] for _ in range(self.B)]
        else:
            self.ids[b], self.keys[b] = [], []

    # ------------------------------------------------------------------
    def _common_prefix(self, b, ids, keys):
        old_i, old_k = self.ids[b], self.keys[b]
        n = min(len(old_i), len(ids) - 1)                      # keep >= 1 new token: its hidden state is the output
        j = 0
        while j < n and old_i[j] == ids[j] and old_k[j] == keys[j] and keys[j] is not False:
            j += 1
        return j

    @torch.no_grad()
    def extend(self, ids_list, vis_idx_list=None, vis_all=None, vis_keys=None, return_rows="last"):
        """Bring the cache to the prompts `ids_list` (B python lists of token ids, no padding) and return the final-norm
        hidden state of each sample's LAST token, [B, d] bf16.

        vis_idx_list[b][j] = row of `vis_all` ([R, d] fp32, device) added to token j's embedding, or -1;
        vis_keys[r] identifies row r across calls: equal keys = the same constant embedding (a `<hist>` token of an
        earlier step), `False` = never reusable (candidates change every step).  Tokens are reused from the cache while
        ids and keys match; the rest is recomputed."""
        m, cfg = self.model, self.model.cfg
        st, dev = m.store, m.device
        B, cap, H, hd, d, eps = self.B, self.cap, cfg.num_heads, cfg.head_dim, cfg.hidden_size, cfg.rms_norm_eps
        assert len(ids_list) == B
        keys_list = []
        for b in range(B):
            L = len(ids_list[b])
            assert 0 < L <= cap, f"prompt of {L} tokens does not fit the cache capacity {cap}"
            if vis_idx_list is None:
                keys_list.append([None] * L)
            else:
                keys_list.append([None if r < 0 else (False if vis_keys is None else vis_keys[r]) for r in vis_idx_list[b]])
        P = [self._common_prefix(b, ids_list[b], keys_list[b]) for b in range(B)]
        n = [len(ids_list[b]) - P[b] for b in range(B)]
        N = max(n)
        M = B * N
        junk = B * cap
        ids_new = torch.full((M,), cfg.pad_token_id, dtype=I32)
        vix_new = torch.full((M,), -1, dtype=I32)
        pos_new = torch.zeros((M,), dtype=I32)
        crow = torch.full((M,), junk, dtype=I32)               # cache row each block row scatters to
        grow = torch.zeros((M,), dtype=I32)                    # cache row each block row gathers its attention output from
        last = torch.zeros((B,), dtype=I32)
        for b in range(B):
            s, e = b * N, b * N + n[b]
            ids_new[s:e] = torch.tensor(ids_list[b][P[b]:], dtype=I32)
            if vis_idx_list is not None:
                vix_new[s:e] = torch.tensor(vis_idx_list[b][P[b]:], dtype=I32)
            ar = torch.arange(P[b], P[b] + n[b], dtype=I32)
            pos_new[s:e] = ar
            crow[s:e] = b * cap + ar
            grow[s:e] = b * cap + ar
            grow[e:b * N + N] = b * cap
            last[b] = e - 1
        ids_d, vix_d, pos_d, crow_d, grow_d, last_d = (ops.h2d(t, dev) for t in (ids_new, vix_new, pos_new, crow, grow, last))
        Lmax = max(len(x) for x in ids_list)
        qmin = (min(P) // 128) * 128

        x = ops.embed_vis(st.p("lang_model.model.embed_tokens.weight"), ids_d, vix_d, vis_all)
        for i in range(cfg.num_layers):
            p = f"lang_model.model.layers.{i}."
            n1, _ = ops.rmsnorm_fwd(x, st.p(p + "input_layernorm.weight"), eps)
            qkv = m.lm_linear(n1, i, "qkv")
            ops.rope_rows_(qkv, m.rope_cos, m.rope_sin, pos_d, H, hd)
            ops.scatter_rows_bf16_(qkv, crow_d, self.qkv[i])
            ops.attn_fwd_strided(self.qkv[i], self.kv0, B, Lmax, cap, H, hd, out=self.attn, lse2=self.lse, q_row_min=qmin)
            attn = ops.gather_rows_bf16(self.attn, grow_d)
            x1 = m.lm_linear(attn, i, "o", R=x, epilogue=ops.EPI_RESID)
            n2, _ = ops.rmsnorm_fwd(x1, st.p(p + "post_attention_layernorm.weight"), eps)
            gu = m.lm_linear(n2, i, "gate_up")
            h = ops.swiglu_fwd(gu)
            x = m.lm_linear(h, i, "down", R=x1, epilogue=ops.EPI_RESID)
        for b in range(B):
            self.ids[b], self.keys[b] = list(ids_list[b]), list(keys_list[b])
        self.last_stats = {"prefix": P, "new": n, "block_rows": M}
        if return_rows == "all":
            Hs, _ = ops.rmsnorm_fwd(x, st.p("lang_model.model.norm.weight"), eps)
            return Hs, (N, n)
        x_last = ops.gather_rows_bf16(x, last_d)
        Hs, _ = ops.rmsnorm_fwd(x_last, st.p("lang_model.model.norm.weight"), eps)
        return Hs

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def logits_last(self, Hs_last):
        """lm_head on [B, d] -> fp32 [B, V] with the special ids at -inf (modified_lm.py:119-124)."""
        cfg, st = self.model.cfg, self.model.store
        lg = ops.gemm_bf16(ops.NT, Hs_last, st.lm_head_padded())[:, :cfg.vocab_size].float()
        lg[:, list(cfg.special_token_ids)] = float("-inf")
        return lg

    @torch.no_grad()
    def generate(self, ids_list, vis_idx_list=None, vis_all=None, max_new_tokens=50, eos_token_id=2, pad_token_id=0,
                 trie=None):
        """Greedy decoding with HF's bookkeeping (`generate(do_sample=False)`): finished rows emit `pad_token_id`, the
        loop ends when every row has produced `eos_token_id` or after `max_new_tokens`.  `trie` (tools/trie.py protocol:
        `.root`, `.get_child_index(node)`, `.get_next_node(node, w)`) constrains each step like TrieLogitsProcessor.
        Returns B lists with the new tokens only."""
        B = self.B
        self.reset()
        seqs = [list(x) for x in ids_list]
        vix = None if vis_idx_list is None else [list(v) for v in vis_idx_list]
        out = [[] for _ in range(B)]
        unfinished = [True] * B
        nodes = [trie.root for _ in range(B)] if trie is not None else None
        keys = None if vis_all is None else [("gen", r) for r in range(vis_all.shape[0])]   # constant within one call
        for step in range(max_new_tokens):
            Hs = self.extend(seqs, vix, vis_all, keys)
            lg = self.logits_last(Hs)
            if trie is not None:
                if step > 0:
                    nodes = [trie.get_next_node(nodes[b], seqs[b][-1]) for b in range(B)]
                allow = torch.zeros(lg.shape, dtype=torch.bool)
                for b in range(B):
                    allow[b, trie.get_child_index(nodes[b])] = True
                lg = lg.masked_fill(ops.h2d(allow.logical_not(), lg.device), float("-inf"))
            nxt = torch.argmax(lg, dim=-1).tolist()
            for b in range(B):
                t = nxt[b] if unfinished[b] else pad_token_id
                out[b].append(t)
                seqs[b].append(t)
                if vix is not None:
                    vix[b].append(-1)
                if unfinished[b] and t == eos_token_id:
                    unfinished[b] = False
            if not any(unfinished):
                break
        return out
