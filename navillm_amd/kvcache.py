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
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from . import ops
from . import debug as _debug

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
DEVICE_GREEDY = os.environ.get("NAVILLM_DEVICE_GREEDY", "1") != "0"      # greedy decoding with the loop on the device (no trie)
USE_HIP_GRAPH = os.environ.get("NAVILLM_DECODE_GRAPH", "1") != "0"       # ... replayed from a captured hipGraph


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
        self._dec, self._dec_key, self._ws = None, None, None
        self._greedy = None                                    # device-side decode loop: buffers + captured hipGraph
        self._key_ids = {}                                     # reuse key (hashable) -> small int
        self.reset()

    def __del__(self):
        d, self._dec = getattr(self, "_dec", None), None
        if d:
            try:
                ops._L().nv_decoder_destroy(d)
            except Exception:
                pass

    def reset(self, b=None):
        """forget the cached tokens (of sample b, or of every sample) -- e.g. at an episode boundary"""
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int64))
        if b is None:
            self.state = [empty for _ in range(self.B)]
            self._key_ids.clear()
            if _debug.POISON:                      # NAVILLM_POISON=1: nothing of the forgotten prompts may be read again
                for t in self.qkv + [self.attn]:
                    _debug.poison_(t, masked_reads=True)
                _debug.poison_(self.lse)
        else:
            self.state[b] = empty

    # compatibility views (tests / callers read the cached ids)
    @property
    def ids(self):
        return [s[0].tolist() for s in self.state]

    # ------------------------------------------------------------------ the native layer loop (navillm_amd/csrc/decoder_runtime.cpp)
    def _decoder(self):
        m, cfg, st = self.model, self.model.cfg, self.model.store
        st.wait_params()           # the native layer loop reads the weights through raw pointers: behind a pending optimizer update
        # keyed by what the layer table POINTS AT, not by the stream (with the stream in the key the decoder was torn
        # down and rebuilt -- ~200 ctypes calls and a fresh split-K workspace allocation + memset -- INSIDE the hipGraph capture of
        # the greedy step, which runs on a side stream).  The split-K workspace handed to the decoder is only touched by the tile
        # GEMMs of prefill-sized steps (M > 16), which always run on the stream the decoder was created on; the captured decode
        # step (M = B <= 16) runs the weight streamers and never reads it.
        key = (id(m.fp8), m.fp8 is not None and m.fp8.resident is not None, m.store.param["lm"].data_ptr(),
               m.fp8 is not None and m.fp8.overlap)
        if self._dec is not None and self._dec_key == key:
            return self._dec
        L = ops._L()
        if self._dec is not None:
            L.nv_decoder_destroy(self._dec)
        dec = ctypes.c_void_p(L.nv_decoder_create(cfg.num_layers, cfg.hidden_size, cfg.num_heads, cfg.head_dim, cfg.intermediate_size,
                                                  cfg.rms_norm_eps))
        if not dec:
            raise _lib.NaviLLMHipError("nv_decoder_create failed")
        kinds = ("qkv", "o", "gate_up", "down")
        for i in range(cfg.num_layers):
            p = f"lang_model.model.layers.{i}."
            _lib.check(L.nv_decoder_set_layer(dec, i, st.p(p + "input_layernorm.weight").data_ptr(),
                                              st.p(p + "post_attention_layernorm.weight").data_ptr(), self.qkv[i].data_ptr()), "nv_decoder_set_layer")
            for k, kind in enumerate(kinds):
                if m.fp8 is not None:
                    wb = m.fp8.resident[i][kind].data_ptr() if m.fp8.resident is not None else None      # bf16(s*q) kept resident
                    rc = L.nv_decoder_set_weight(dec, i, k, wb, m.fp8.codes[i][kind].data_ptr(), m.fp8.scales[i][kind].data_ptr())
                else:
                    rc = L.nv_decoder_set_weight(dec, i, k, m.lm_w(i, kind).data_ptr(), None, None)
                _lib.check(rc, "nv_decoder_set_weight")
        scratch = m.fp8._scratch.data_ptr() if (m.fp8 is not None and m.fp8._scratch is not None) else None
        if m.fp8 is not None:
            _lib.check(L.nv_decoder_set_fp8_gemm_mode(dec, int(m.fp8.gemm_mode)), "nv_decoder_set_fp8_gemm_mode")
        _lib.check(L.nv_decoder_set_shared(dec, m.rope_cos.data_ptr(), m.rope_sin.data_ptr(), st.p("lang_model.model.norm.weight").data_ptr(),
                                           ops._gemm_ws(m.device) if ops.SPLITK_TAIL else None, scratch), "nv_decoder_set_shared")
        if m.fp8 is not None and m.fp8.overlap:
            pa, pb, side = m.fp8.pipe()
            _lib.check(L.nv_decoder_set_fp8_overlap(dec, pa.data_ptr(), pb.data_ptr(), side.cuda_stream), "nv_decoder_set_fp8_overlap")
        self._dec, self._dec_key = dec, key
        return dec

    def _key_codes(self, vis_idx, vis_keys):
        """per-token reuse code: 0 = plain token, -1 = visual row that must be recomputed, > 0 = interned key of a constant row"""
        vi = np.asarray(vis_idx, dtype=np.int64)
        codes = np.zeros(vi.size, dtype=np.int64)
        for j in np.flatnonzero(vi >= 0):                  # the few visual tokens only (a Python loop over all ~650 tokens of all
            k = False if vis_keys is None else vis_keys[int(vi[j])]     # 8 prompts was 1 ms of host time per step)
            codes[j] = -1 if k is False else self._key_ids.setdefault(k, len(self._key_ids) + 1)
        return codes

    @torch.no_grad()
    def extend(self, ids_list, vis_idx_list=None, vis_all=None, vis_keys=None, return_rows="last"):
        """Bring the cache to the prompts `ids_list` (B python lists of token ids, no padding) and return the final-norm
        hidden state of each sample's LAST token, [B, d] bf16.

        (return_rows="all": the hidden states of all new rows, packed sample after sample, with their offsets.)
        vis_idx_list[b][j] = row of `vis_all` ([R, d] fp32, device) added to token j's embedding, or -1;
        vis_keys[r] identifies row r across calls: equal keys = the same constant embedding (a `<hist>` token of an
        earlier step), `False` = never reusable (candidates change every step).  Tokens are reused from the cache while
        ids and keys match; the rest is recomputed.  The decoder layers run as ONE native call (nv_decoder_extend)."""
        m, cfg = self.model, self.model.cfg
        st, dev = m.store, m.device
        B, cap, H, d = self.B, self.cap, cfg.num_heads, cfg.hidden_size
        assert len(ids_list) == B
        new_state, P, n = [], [], []
        for b in range(B):
            ids = np.asarray(ids_list[b], dtype=np.int64)
            Lb = ids.size
            assert 0 < Lb <= cap, f"prompt of {Lb} tokens does not fit the cache capacity {cap}"
            codes = np.zeros(Lb, np.int64) if vis_idx_list is None else self._key_codes(vis_idx_list[b], vis_keys)
            old_i, old_c = self.state[b]
            k = min(old_i.size, Lb - 1)                        # keep >= 1 new token: its hidden state is the output
            diff = np.flatnonzero((old_i[:k] != ids[:k]) | (old_c[:k] != codes[:k]) | (codes[:k] < 0))
            P.append(int(diff[0]) if diff.size else k)
            n.append(Lb - P[-1])
            new_state.append((ids, codes))
        # the new rows, PACKED sample after sample (no padding to the longest suffix: every kernel of the step is row-wise, and at
        # ~100 new tokens per sample the padded block layout cost a whole extra 256-row GEMM tile row on a typical step)
        off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        M = int(off[-1])
        ids_new = np.empty(M, np.int32)
        vix_new = np.full(M, -1, np.int32)
        pos_new = np.empty(M, np.int32)
        crow = np.empty(M, np.int32)                           # cache row each new row scatters to ...
        last = np.zeros(B, np.int32)
        for b in range(B):
            s, e = int(off[b]), int(off[b + 1])
            ids_new[s:e] = new_state[b][0][P[b]:]
            if vis_idx_list is not None:
                vix_new[s:e] = vis_idx_list[b][P[b]:]
            ar = np.arange(P[b], P[b] + n[b], dtype=np.int32)
            pos_new[s:e] = ar
            crow[s:e] = b * cap + ar
            last[b] = e - 1
        grow = crow                                            # ... and reads its attention output back from
        # one pinned staging buffer, one H2D copy for all six index arrays
        packed = np.concatenate([ids_new, vix_new, pos_new, crow, grow, last])
        idx = ops.h2d(torch.from_numpy(packed), dev)
        ids_d, vix_d, pos_d, crow_d, grow_d = (idx[k * M:(k + 1) * M] for k in range(5))
        last_d = idx[5 * M:5 * M + B]
        Lmax = max(s[0].size for s in new_state)
        qmin = (min(P) // 128) * 128

        x = ops.embed_vis(st.p("lang_model.model.embed_tokens.weight"), ids_d, vix_d, vis_all)
        dec = self._decoder()
        L = ops._L()
        need = L.nv_decoder_workspace_bytes(dec, M)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        want_all = return_rows == "all"
        hs = torch.empty((B, d), dtype=BF16, device=dev)
        hs_all = torch.empty((M, d), dtype=BF16, device=dev) if want_all else None
        rc = L.nv_decoder_extend(dec, x.data_ptr(), pos_d.data_ptr(), crow_d.data_ptr(), grow_d.data_ptr(), self.kv0.data_ptr(),
                                 self.attn.data_ptr(), self.lse.data_ptr(), last_d.data_ptr(), hs.data_ptr(), ops._p(hs_all), M, B, Lmax, cap,
                                 qmin, None, self._ws.data_ptr(), self._ws.numel(), ops._st())
        _lib.check(rc, "nv_decoder_extend")
        if m.fp8 is not None and M > 16:
            m.fp8.invalidate()             # the layer loop's overlapped pre-pass wrote the two de-quantisation panels
        self.state = new_state
        self.last_stats = {"prefix": P, "new": n, "block_rows": M}
        if want_all:
            return hs_all, (off, n)
        return hs

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
                 trie=None, do_sample=False, temperature=1.0, top_k=50):
        """Greedy decoding with HF's bookkeeping (`generate(do_sample=False)`): finished rows emit `pad_token_id`, the
        loop ends when every row has produced `eos_token_id` or after `max_new_tokens`.  `trie` (tools/trie.py protocol:
        `.root`, `.get_child_index(node)`, `.get_next_node(node, w)`) constrains each step like TrieLogitsProcessor.
        do_sample=True (tasks/agents/llava.py:58-62 forwards `--do_sample --temperature`): HF's `sample()` -- after the mask and
        the trie, logits / temperature, the default top-k (50) warper, softmax, one multinomial draw per row from torch's CUDA
        generator (same distribution as the reference; the streams of two RNG implementations cannot be compared draw by draw).
        Returns B lists with the new tokens only."""
        B = self.B
        self.reset()
        seqs = [list(x) for x in ids_list]
        vix = None if vis_idx_list is None else [list(v) for v in vis_idx_list]
        keys = None if vis_all is None else [("gen", r) for r in range(vis_all.shape[0])]   # constant within one call
        if trie is None and not do_sample and DEVICE_GREEDY and max_new_tokens > 0 and max(len(x) for x in seqs) + max_new_tokens <= self.cap:
            return self._generate_on_device(seqs, vix, vis_all, keys, max_new_tokens, eos_token_id, pad_token_id)
        out = [[] for _ in range(B)]
        unfinished = [True] * B
        nodes = [trie.root for _ in range(B)] if trie is not None else None
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
            if do_sample:
                if temperature != 1.0:
                    lg = lg / temperature
                if top_k and top_k < lg.shape[-1]:
                    kth = torch.topk(lg, top_k, dim=-1).values[:, -1:]
                    lg = lg.masked_fill(lg < kth, float("-inf"))
                nxt = torch.multinomial(torch.softmax(lg, dim=-1), 1).view(-1).tolist()
            else:
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

    # ------------------------------------------------------------------ greedy decoding with the loop on the device
    def _greedy_state(self, max_steps):
        """persistent buffers of the device loop (fixed addresses: the step is replayed from a hipGraph)"""
        g = self._greedy
        if g is not None and g["max_steps"] >= max_steps:
            return g
        m, cfg, dev, B = self.model, self.model.cfg, self.model.device, self.B
        L = ops._L()
        n = L.nv_decode_state_ints(B)
        vp = m.store.vocab_pad
        g = {"max_steps": max(max_steps, 64), "graph": None, "graph_key": None,
             "state": torch.zeros((n,), dtype=I32, device=dev), "hs": torch.zeros((B, cfg.hidden_size), dtype=BF16, device=dev),
             "x": torch.zeros((B, cfg.hidden_size), dtype=BF16, device=dev), "logits": torch.zeros((B, vp), dtype=BF16, device=dev),
             "fin_host": [torch.zeros((B,), dtype=I32).pin_memory() for _ in range(3)]}     # landing buffers of the lagging `fin` poll
        g["out"] = torch.zeros((g["max_steps"], B), dtype=I32, device=dev)
        self._greedy = g
        return g

    def _greedy_step(self, g, eos, pad, stream, dec):
        """`dec`: the native decoder, resolved by the caller BEFORE any stream capture (nothing here may allocate or rebuild)"""
        m, cfg, st = self.model, self.model.cfg, self.model.store
        sp = cfg.special_token_ids
        rc = ops._L().nv_decoder_greedy_step(dec, g["hs"].data_ptr(), st.p("lang_model.model.embed_tokens.weight").data_ptr(),
                                             st.lm_head_padded().data_ptr(), st.vocab_pad, cfg.vocab_size, sp[0], len(sp), g["logits"].data_ptr(),
                                             g["x"].data_ptr(), g["state"].data_ptr(), g["out"].data_ptr(), g["max_steps"], self.kv0.data_ptr(),
                                             self.attn.data_ptr(), self.lse.data_ptr(), self.B, self.cap, eos, pad, self._ws.data_ptr(),
                                             self._ws.numel(), stream)
        _lib.check(rc, "nv_decoder_greedy_step")

    @torch.no_grad()
    def _generate_on_device(self, seqs, vix, vis_all, keys, max_new_tokens, eos, pad):
        """prefill through `extend` (host-built indices, once), then max_new_tokens replays of ONE captured step
        (nv_decoder_greedy_step: lm_head -> masked argmax + finished/pad bookkeeping -> cache indices -> embedding -> decoder layers).
        The host never waits for a token: it polls the `fin` flags two steps behind and trims the output where HF would have stopped."""
        B, dev = self.B, self.model.device
        g = self._greedy_state(max_new_tokens)
        L = ops._L()
        dec = self._decoder()
        need = L.nv_decoder_workspace_bytes(dec, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        if USE_HIP_GRAPH and g["graph"] is None:
            # one eager step on a scratch state BEFORE the prefill: the kernels' first-use initialisation (hipFuncSetAttribute, device
            # queries) must not happen inside a stream capture.  It writes cache row 0 of every sample, which the prefill rewrites.
            g["state"].zero_()
            self._greedy_step(g, eos, pad, ops._st(), dec)
        Hs = self.extend(seqs, vix, vis_all, keys)
        assert self._decoder() is dec, "the decoder table changed between prefill and decode"
        n = g["state"].numel()
        init = np.zeros(n, np.int32)
        init[2 * B:3 * B] = [len(x) for x in seqs]
        g["state"].copy_(ops.h2d(torch.from_numpy(init), dev))
        g["hs"].copy_(Hs)
        key = (self._dec_key, self._ws.data_ptr(), eos, pad, g["out"].data_ptr())
        if USE_HIP_GRAPH and g["graph_key"] != key:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                self._greedy_step(g, eos, pad, torch.cuda.current_stream().cuda_stream, dec)
            g["graph"], g["graph_key"] = graph, key            # capture enqueues nothing: the first replay is step 0
        events = []
        steps = 0
        for t in range(max_new_tokens):
            if USE_HIP_GRAPH:
                g["graph"].replay()
            else:
                self._greedy_step(g, eos, pad, ops._st(), dec)
            steps += 1
            if len(events) >= 2:                               # flags as of two steps ago: the host stays ahead of the GPU
                ev, snap = events.pop(0)
                ev.synchronize()
                if bool(snap.all()):
                    break
            snap = g["fin_host"][t % 3]                        # at most two polls are outstanding
            snap.copy_(g["state"][B:2 * B], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events.append((ev, snap))
        toks = g["out"][:steps].cpu().numpy()                  # [steps, B]; syncs
        if int(g["state"][7 * B + 3].item()):
            raise RuntimeError("KVCacheLM.generate: a sample's K/V cache filled up during decoding (prompt length + max_new_tokens > capacity)")
        # HF stops after the step in which the last unfinished row emitted eos: trim the steps the lagging poll let through
        fin = np.zeros(B, bool)
        keep = steps
        for t in range(steps):
            fin |= toks[t] == eos
            if fin.all():
                keep = t + 1
                break
        toks = toks[:keep]
        lens = g["state"][2 * B:3 * B].cpu().numpy()
        # the cache now also holds the generated tokens (all `steps` of them): record it so a later extend() reuses / overwrites correctly
        full = g["out"][:steps].cpu().numpy()
        self.state = [(np.concatenate([np.asarray(seqs[b], np.int64), full[:, b].astype(np.int64)])[:int(lens[b])],
                       np.concatenate([self.state[b][1], np.zeros(steps, np.int64)])[:int(lens[b])]) for b in range(B)]
        return [toks[:, b].tolist() for b in range(B)]
