"""Flat parameter / gradient storage in HBM.

MI355X-first layout decision: all LM-dtype tensors live in ONE contiguous bf16 buffer and all
fp32 tensors in ONE fp32 buffer (same for gradients and both AdamW moments).  Consequences:
  * q|k|v and gate|up weights are adjacent, so each pair is ONE packed GEMM operand;
  * weight-gradient GEMMs accumulate straight into their slice of the flat grad buffer
    (fused `grad += dW` epilogue) -- no per-parameter `.grad` tensors to add later;
  * the optimizer is a single fused kernel over each buffer (K13), and data-parallel
    gradient exchange is an all-reduce over contiguous slices (one per decoder layer),
    with no bucket copies.
`nn.Parameter`s named exactly like the reference's state_dict (navillm_amd/params.py) are views
into these buffers, so `state_dict()/load_state_dict()` keep working (tools/optims.py:12-24,65-78).
"""
import torch
from .params import param_specs

ALIGN = 64  # elements; keeps every tensor 128-B aligned for 16-B vector access and DMA


def _lm_order(cfg):
    """bf16 group order: per layer q,k,v | o | gate,up | down | norms, so packed views exist."""
    names = ["lang_model.model.embed_tokens.weight"]
    for i in range(cfg.num_layers):
        p = f"lang_model.model.layers.{i}."
        names += [p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                  p + "self_attn.o_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight",
                  p + "mlp.down_proj.weight", p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]
    names += ["lang_model.model.norm.weight", "lang_model.lm_head.weight"]
    return names


class FlatStore:
    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = torch.device(device)
        specs = param_specs(cfg)
        shape_of = {n: s for n, s, _ in specs}
        group_of = {n: g for n, _, g in specs}
        lm_names = [n for n in _lm_order(cfg)] + [n for n, _, g in specs if g == "lm" and not n.startswith("lang_model.")]
        assert set(lm_names) == {n for n, _, g in specs if g == "lm"}
        f32_names = [n for n, _, g in specs if g == "f32"]
        self.lm_dtype = torch.bfloat16 if cfg.lm_is_bf16 else torch.float32
        self.offsets = {}
        self.sizes = {}
        self.alloc_sizes = {}     # size incl. alignment padding: offsets[n] + alloc_sizes[n] == offset of the next tensor
        self.layers_zero = False  # True from a zero_grad() (the buffers hold zeros) until a backward writes DECODER-LAYER gradients: the first
                                  # writer may then STORE instead of read-accumulate (navillm_amd/episode.py: the episode's wgrad GEMMs)
        self.grad_writes = 0      # bumped whenever a backward writes (or is about to write) into .grad -- see FlatAdamW.zero_grad
        self.tainted = None       # set by NavModel.episode_abort() when the aborted episode had already written PART of its gradients
        self.touched = set()      # names that have received a gradient at least once (navillm_amd/optim.py: which tensors AdamW updates)
        self.names = {"lm": lm_names, "f32": f32_names}
        self.shape_of = shape_of
        self.group_of = group_of
        self.vocab_pad = (cfg.vocab_size + 63) // 64 * 64
        tot = {}
        for grp, names in self.names.items():
            off = 0
            for n in names:
                numel = 1
                for s in shape_of[n]:
                    numel *= s
                self.offsets[n] = off
                self.sizes[n] = numel
                alloc = numel
                if n == "lang_model.lm_head.weight":
                    # rows padded to a multiple of 64 (zero, never trained) so the vocab dimension can be
                    # the contraction of the dgrad GEMM (K % 64 == 0) without copying logits
                    alloc = self.vocab_pad * shape_of[n][1]
                # packed partners must stay adjacent: q/k/v and gate/up sizes are multiples of ALIGN
                self.alloc_sizes[n] = (alloc + ALIGN - 1) // ALIGN * ALIGN
                off += self.alloc_sizes[n]
            tot[grp] = off
        self.total = tot
        dt = {"lm": self.lm_dtype, "f32": torch.float32}
        self.param = {g: torch.zeros(tot[g], dtype=dt[g], device=self.device) for g in tot}
        self.grad = {g: torch.zeros(tot[g], dtype=dt[g], device=self.device) for g in tot}
        self.exp_avg = None
        self.exp_avg_sq = None
        self._upd = None          # an AdamW update of the LM group still running on the optimizer's side stream (FlatAdamW.step, round 6)

    # ---- asynchronous update of the LM group (navillm_amd/optim.py::FlatAdamW.step with overlap): the update kernels run on a side stream,
    # decoder layer by decoder layer in flat order, while the launch stream goes on with the next episode's scene-encoder / fusion steps
    # (fp32 group: updated on the launch stream itself).  Whoever touches LM-group parameters OR gradients next waits for exactly the
    # part it needs: every accessor below (`p`, `g`, `qkv`, `gate_up`, `lm_head_padded`) and `PrefixEpisode._weights` calls
    # `wait_params`; code that reads the raw buffers (`state_dict`, the native decoder's pointer table, the DP broadcast, the next clip /
    # step, a full zero_grad) calls it without arguments = waits for all of it.  Other streams (the DP exchange, the wgrad side stream)
    # fork from the launch stream AFTER it has waited, so they inherit the order.
    def begin_async_update(self, stream, head_ev, layer_evs, done_ev, home):
        self._upd = {"stream": stream, "head": head_ev, "layer": layer_evs, "done": done_ev, "home": home, "waited": set()}

    def wait_params(self, name=None, layer=None):
        """order the CURRENT stream after the pending update of `name` / decoder layer `layer` (None, None: all of the LM group)"""
        u = self._upd
        if u is None:
            return
        import torch
        if layer is None and name is not None:
            if name.startswith("lang_model.model.layers."):
                layer = int(name.split(".")[3])
            elif name != "lang_model.model.embed_tokens.weight":
                name = None                                  # final norm, lm_head, the heads: the tail of the buffer = the whole update
        key = ("L", layer) if layer is not None else (("H",) if name is not None else ("D",))
        cur = torch.cuda.current_stream(self.device)
        at_home = cur.cuda_stream == u["home"]
        if at_home and (key in u["waited"] or ("D",) in u["waited"]):
            return
        if cur.cuda_stream == u["stream"].cuda_stream:
            return
        ev = u["layer"][layer] if layer is not None else (u["head"] if name is not None else u["done"])
        cur.wait_event(ev)
        if at_home:
            u["waited"].add(key)
            if key == ("D",):
                self._upd = None                             # the launch stream is behind the whole update: nothing left to order

    # ---- views
    def _view(self, store, name):
        g = self.group_of[name]
        o, n = self.offsets[name], self.sizes[name]
        return store[g][o:o + n].view(self.shape_of[name])

    def p(self, name):
        if name in getattr(self, "released", ()):
            raise RuntimeError(f"{name}: released by to_fp8_weight_only() (weight-only fp8 deployment); use NavModel.lm_w / lm_linear")
        if self._upd is not None and self.group_of[name] == "lm":
            self.wait_params(name)
        return self._view(self.param, name)

    def g(self, name):
        if self.grad is None:
            raise RuntimeError("gradient buffers were released by to_fp8_weight_only(): the fp8 deployment is inference only")
        if self._upd is not None and self.group_of[name] == "lm":
            self.wait_params(name)                 # (the update zeroes the gradient it consumes: writers are ordered behind it too)
        return self._view(self.grad, name)

    def _packed(self, store, first, last):
        """[rows_total, cols] view spanning adjacent tensors first..last (same cols)."""
        if first in getattr(self, "released", ()):
            raise RuntimeError(f"{first}: the bf16 decoder weights were released by to_fp8_weight_only(); use NavModel.lm_w / lm_linear")
        cols = self.shape_of[first][1]
        o0 = self.offsets[first]
        o1 = self.offsets[last] + self.sizes[last]
        assert (o1 - o0) % cols == 0
        v = store["lm"][o0:o1].view(-1, cols)
        return v

    def qkv(self, i, grad=False):
        p = f"lang_model.model.layers.{i}.self_attn."
        if self._upd is not None:
            self.wait_params(layer=i)
        v = self._packed(self.grad if grad else self.param, p + "q_proj.weight", p + "v_proj.weight")
        assert v.shape[0] == 3 * self.cfg.hidden_size, "q/k/v must be adjacent and unpadded"
        return v

    def gate_up(self, i, grad=False):
        p = f"lang_model.model.layers.{i}.mlp."
        if self._upd is not None:
            self.wait_params(layer=i)
        v = self._packed(self.grad if grad else self.param, p + "gate_proj.weight", p + "up_proj.weight")
        assert v.shape[0] == 2 * self.cfg.intermediate_size, "gate/up must be adjacent and unpadded"
        return v

    def lm_head_padded(self, grad=False):
        o = self.offsets["lang_model.lm_head.weight"]
        d = self.cfg.hidden_size
        if self._upd is not None:
            self.wait_params()
        return (self.grad if grad else self.param)["lm"][o:o + self.vocab_pad * d].view(self.vocab_pad, d)

    def layer_slice(self, i):
        """(start, end) element range of decoder layer i inside the lm buffers (one DP bucket)."""
        p = f"lang_model.model.layers.{i}."
        s = self.offsets[p + "self_attn.q_proj.weight"]
        last = p + "post_attention_layernorm.weight"
        e = self.offsets[last] + (self.sizes[last] + ALIGN - 1) // ALIGN * ALIGN
        return s, e

    # ---- "this tensor has a gradient now" (torch: p.grad is no longer None); reported by the backward functions
    def touch(self, *names):
        if self.layers_zero and any(n.startswith("lang_model.model.layers.") for n in names):
            self.layers_zero = False
        self.grad_writes += 1           # a gradient was (or is about to be) written: FlatAdamW's fused zero-grad bookkeeping is stale
        self.touched.update(names)

    def assert_layers_zero(self, where):
        """debug (NAVILLM_POISON=1): a first-writer STORE of the decoder layers' weight gradients is only correct while those
        buffers really hold zeros -- every writer must have gone through touch() / touch_layers(), which clear `layers_zero`"""
        import torch
        s0, _ = self.layer_slice(0)
        _, e1 = self.layer_slice(self.cfg.num_layers - 1)
        nz = int(torch.count_nonzero(self.grad["lm"][s0:e1]))
        if nz:
            raise RuntimeError(f"{where}: FlatStore.layers_zero is set but {nz} decoder-layer gradient elements are non-zero -- a writer "
                               "bypassed FlatStore.touch() / touch_layers() (a STORE epilogue would have dropped its gradient)")

    def touch_layers(self):
        """every decoder-layer tensor + the final norm (LlamaStack.backward accumulates into all of them)"""
        self.layers_zero = False
        self.grad_writes += 1
        if "lang_model.model.norm.weight" not in self.touched:
            self.touched.update(n for n in self.names["lm"] if n.startswith("lang_model.model.layers.") or n == "lang_model.model.norm.weight")

    def zero_grad(self):
        self.tainted = None
        self.wait_params()
        if self.grad is not None:
            for g in self.grad.values():
                g.zero_()
            self.layers_zero = True

    def release_decoder_layers_and_grads(self, named_params):
        """Inference deployment with weight-only fp8 decoder weights (navillm_amd/fp8.py): drop the bf16 copies of the decoder's
        Linear weights and every gradient buffer.  The remaining LM-dtype tensors (embeddings, norms, lm_head, heads) move to
        a compact buffer; the nn.Parameters are re-pointed at it (released ones become empty)."""
        self.wait_params()
        gone = lambda n: n.startswith("lang_model.model.layers.") and n.endswith("_proj.weight")
        keep = [n for n in self.names["lm"] if not gone(n)]
        off, new_off = 0, {}
        for n in keep:
            new_off[n] = off
            off += self.alloc_sizes[n]
        new = torch.zeros(off, dtype=self.lm_dtype, device=self.device)
        for n in keep:
            new[new_off[n]:new_off[n] + self.alloc_sizes[n]].copy_(self.param["lm"][self.offsets[n]:self.offsets[n] + self.alloc_sizes[n]])
        self.param["lm"] = new
        self.offsets.update(new_off)
        self.released = {n for n in self.names["lm"] if gone(n)}
        self.names["lm"] = keep
        self.total["lm"] = off
        self.grad = None
        self.exp_avg = self.exp_avg_sq = None
        for n, p in named_params.items():
            p.grad = None
            p.requires_grad_(False)
            p.data = torch.empty(0, dtype=p.dtype, device=self.device) if n in self.released else self.p(n)

    def release_grads(self, named_params):
        """inference deployment that keeps the parameters: drop the gradient buffers and the optimizer state only"""
        self.grad = None
        self.exp_avg = self.exp_avg_sq = None
        for p in named_params.values():
            p.grad = None
            p.requires_grad_(False)

    def init_optimizer_state(self):
        if self.exp_avg is None:
            self.exp_avg = {g: torch.zeros_like(t) for g, t in self.param.items()}
            self.exp_avg_sq = {g: torch.zeros_like(t) for g, t in self.param.items()}

    def load_state_dict_tensors(self, sd):
        self.wait_params()
        for n in self.offsets:
            if n in sd:
                self.p(n).copy_(sd[n].to(self.p(n).dtype))
