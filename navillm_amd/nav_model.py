"""Drop-in mirror of the reference's `NavModel` (models/nav_model.py:33-451) for MI355X.

Same constructor (`NavModel(args, logger, model_config)`), same call protocol
(`model(mode, batch, **kw) -> dict`, modes of nav_model.py:96-126), same attributes the agents
touch (`.lang_model.cls_token`, `.lang_model.tokenizer`, `.lang_model.cls_token_id`), same
`state_dict()` key layout.  Underneath, every device computation is a libnavillm_hip.so launch
(navillm_amd/functions.py); the host only does what the reference also does on the host:
tokenisation, vpid matching (nav_model.py:174-190), candidate permutation (:216-223).

Inference extras: `enable_kv_cache()` = prompt-prefix K/V reuse across no-grad navigation steps;
`training=False` in the LM-loss modes = greedy generation (navillm_amd/kvcache.py).
Not mirrored (raise NotImplementedError): fp32 LM, OPT LMs.
"""
import collections
import contextlib
import functools
import itertools
import os

import numpy as np
import torch
import torch.nn as nn

from . import functions as Fn
from . import graph
from . import ops
from .config import NavConfig
from .flat import FlatStore
from .params import param_specs, synth_tensor

F32, BF16 = torch.float32, torch.bfloat16


class _Box(nn.Module):
    """anonymous container so parameters get the reference's dotted names"""


class LangModelShell(nn.Module):
    """Carries what the agents read from `model.lang_model` (modified_lm.py:56-87)."""

    def __init__(self, cfg, tokenizer=None):
        super().__init__()
        self.cand_token = ["<cand>"]
        self.hist_token = ["<hist>"]
        self.obj_token = ["<obj>"]
        self.cls_token = ["<cls_1>", "<cls_2>"]
        self.cand_token_id = [cfg.cand_token_id]
        self.hist_token_id = [cfg.hist_token_id]
        self.obj_token_id = [cfg.obj_token_id]
        self.cls_token_id = list(cfg.cls_token_ids)
        self.special_token_ids = list(cfg.special_token_ids)
        self.hidden_size = cfg.hidden_size
        self.model_type = BF16 if cfg.lm_is_bf16 else F32
        self.tokenizer = tokenizer
        self.max_length = 1024          # modified_lm.py:57,77-87: prompts are LEFT-truncated at this many tokens

    def tokenize(self, text, add_special_tokens=True):
        """modified_lm.py:77-87 (left pad, left truncation at 1024, token_type_ids)."""
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer attached: pass pre-tokenised batch['input_ids'] / "
                               "batch['attention_mask'] or build the model from a tokenizer directory")
        return self.tokenizer(text, max_length=self.max_length, padding=True, truncation=True, return_tensors="pt",
                              add_special_tokens=add_special_tokens, return_token_type_ids=True)


def load_tokenizer(path, cfg):
    """LlamaTokenizer + the five special tokens + <PAD> exactly as init_tokenizer does (modified_lm.py:56-75)."""
    from transformers import LlamaTokenizer
    tok = LlamaTokenizer.from_pretrained(path, padding_side="left", truncation_side="left")
    tok.add_special_tokens({"additional_special_tokens": ["<cand>", "<hist>", "<obj>", "<cls_1>", "<cls_2>"]})
    if tok.pad_token is None:
        tok.add_special_tokens({"pad_token": "<PAD>"})
    assert tok.encode("<cand>", add_special_tokens=False) == [cfg.cand_token_id]
    assert tok.encode("<cls_1>", add_special_tokens=False) == [cfg.cls_token_ids[0]]
    assert len(tok) == cfg.vocab_size, (len(tok), cfg.vocab_size)
    return tok


class _Out(dict):
    """dict that also answers `.loss` (llava.py:38 reads `model("3dqa", batch).loss`)."""

    @property
    def loss(self):
        return self["loss"]


class NavModel(nn.Module):
    def __init__(self, args=None, logger=None, model_config=None, *, nav_config=None, device=None, tokenizer=None,
                 seed=0, init=None):
        super().__init__()
        if nav_config is None:
            nav_config = self._config_from_args(args, model_config)
        cfg = self.cfg = nav_config
        self.args = args
        if not cfg.lm_is_bf16:
            raise NotImplementedError("the MI355X path implements the reference's amp_bf16 mode (bf16 LM + heads, fp32 "
                                      "encoder); precision='fp32' has no HIP LM")
        if cfg.head_dim != 128:
            raise NotImplementedError("attention kernels are built for head_dim 128 (Llama/Vicuna 7B/13B)")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("NavModel needs a GPU: there is no CPU fallback (the CPU oracle is test-only)")
        ops._L()  # fail loudly, now, if libnavillm_hip.so is missing
        self.store = FlatStore(cfg, self.device)
        self.hidden_size = cfg.hidden_size
        self.model_type = BF16

        # ---- parameters: nn.Parameters that are VIEWS into the flat buffers, reference names
        if tokenizer is None and args is not None and getattr(args, "pretrained_model_name_or_path", None):
            pth = args.pretrained_model_name_or_path
            if os.path.exists(os.path.join(pth, "tokenizer.model")):
                tokenizer = load_tokenizer(pth, cfg)
        self.lang_model = LangModelShell(cfg, tokenizer)
        self._named = collections.OrderedDict()
        for name, shape, group in param_specs(cfg):
            p = nn.Parameter(self.store.p(name), requires_grad=True)
            p.grad = self.store.g(name)
            p._nv_touch = functools.partial(self.store.touch, name)
            self._attach(name, p)
            self._named[name] = p
        # weights.  nav_config path (benchmarks, tests): seeded synthetic.  args path (the reference's constructor call,
        # train.py:227): what the reference does -- pretrained LM via the HF checkpoint unless from_scratch / resume, torch
        # default inits for everything else; never a silent synthetic fallback (navillm_amd/checkpoint.py)
        if init is None:
            init = "synthetic" if args is None else "reference"
        if init == "synthetic":
            self.init_synthetic(seed)
        elif init == "reference":
            from . import checkpoint as ck
            scratch = getattr(args, "resume_from_checkpoint", None) is not None or bool(getattr(args, "from_scratch", False))
            ck.init_reference_scratch(self, seed, lm=scratch, rest=True)
            if not scratch:
                n = ck.load_hf_llama(self, args.pretrained_model_name_or_path, seed)
                if logger is not None:
                    logger.info(f"loaded {n} LM tensors from {args.pretrained_model_name_or_path}")
            elif logger is not None:
                logger.info("Initialize the model from config.")
        elif init != "none":
            raise ValueError(f"init={init!r}")

        # ---- RoPE tables, built the HF way (fp32 angles on the host, then cast to the LM dtype)
        hd = cfg.head_dim
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
        fr = torch.outer(torch.arange(2048).float(), inv)
        emb = torch.cat([fr, fr], -1)
        self.rope_cos = emb.cos().to(BF16).to(self.device).contiguous()
        self.rope_sin = emb.sin().to(BF16).to(self.device).contiguous()
        self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
        self.arena = Fn.ActivationArena(cfg, self.device)
        self.overlap_wgrad = int(os.environ.get("NAVILLM_OVERLAP_WGRAD", "0") or 0)   # wgrad GEMMs on a side stream (A/B knob)
        self.prune_last_layer = True     # navigation/grounding: last decoder layer computed for the <cls_1> rows only
        self.pack_rows = os.environ.get("NAVILLM_PACK_ROWS", "1") != "0"   # LM over the real tokens only (no left-padding rows)
        self._row_map = None
        self.episode = None              # PrefixEpisode (begin_episode): static prompt prefix computed once per training episode
        self._window = None              # the PrefixEpisode of the open accumulation window (begin_episode(..., accumulate=n))
        self._episode_plain = None       # ... and the non-windowed one it displaced from `self.episode`, kept for the next plain episode
        # round 6: AUTOMATIC episodes -- the unmodified rollout (mp3d_agent.py:660-778) never calls begin_episode / finish_episode; a
        # grad-enabled training-mode navigation call then opens the episode itself and whoever needs `.grad` closes it (`_auto_*` below)
        self.auto_episode = os.environ.get("NAVILLM_AUTO_EPISODE", "1") != "0"
        # ... in which form: "lazy" (default) = the steps' LM forward is deferred too and `fuse_logits` is a losses.LazyLogits handle that
        # forces the pending steps through the decoder the moment something reads it (a teacher-forced rollout never does: its forward
        # runs as one batch, like an explicit teacher_forced=True episode); "step" (NAVILLM_AUTO_EPISODE=step) = every step's forward runs
        # when it is called, `fuse_logits` is a plain tensor
        self.auto_form = "step" if os.environ.get("NAVILLM_AUTO_EPISODE", "1") == "step" else "lazy"
        self._auto_open = False          # the open episode was opened by forward_navigation, not by the caller
        self.auto_stats = {"opened": 0, "closed_by": {}}
        self.fp8 = None                  # Fp8DecoderWeights after to_fp8_weight_only()
        self._gen_kv = None              # K/V cache object kept between generate() calls
        self.flop_log = None             # bench: list of ("lm", tokens, sum of S_b^2, backward?) / ("lm_head", rows, backward?) per LM call
        self.attn_hf_rounding = False    # tests only: attention forward through the parity instrument nv_attn_fwd_hfround_bf16
        self.rope_frame = "batch"        # tests only: "sample" = RoPE positions from each sample's first token in the packed _lm path
        self.kv = None                   # KVCacheLM (enable_kv_cache): prefix reuse across no-grad navigation steps + generation
        self._wgrad_stream = None
        self._dp = None
        self.drop_env_p = cfg.feat_dropout
        self.injected_dropout = None   # tests: dict of keep masks

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def _config_from_args(args, model_config):
        over = dict(image_feat_size=args.image_feat_size, angle_feat_size=args.angle_feat_size,
                    obj_feat_size=args.obj_feat_size, enable_og=bool(args.enable_og), fuse_obj=bool(args.fuse_obj),
                    feat_dropout=args.feat_dropout, precision=args.precision,
                    num_pano_layers=model_config.num_pano_layers)
        return NavConfig.from_hf_dir(args.pretrained_model_name_or_path, **over)

    def _attach(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            if not hasattr(mod, part):
                mod.add_module(part, _Box())
            mod = getattr(mod, part)
        mod.register_parameter(parts[-1], param)

    @torch.no_grad()
    def init_synthetic(self, seed=0):
        """Seeded random weights (navillm_amd/params.py: per-name generators), generated on the device."""
        for name, shape, group in param_specs(self.cfg):
            self.store.p(name).copy_(synth_tensor(name, shape, seed, device="cpu" if self.cfg.hidden_size <= 1024 else self.device))

    @torch.no_grad()
    def load_reference_state_dict(self, sd):
        """Load a NaviLLM checkpoint's `model_state_dict` (tools/optims.py:12-24 semantics: strip `module.`,
        skip shape mismatches)."""
        n = 0
        self.store.wait_params()
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            if k in self._named and tuple(v.shape) == tuple(self._named[k].shape):
                self._named[k].copy_(v.to(self._named[k].dtype))
                n += 1
        return n

    # ---- the decoder's four packed GEMM operands per layer: bf16 views of the flat store, or their weight-only fp8 form
    _LM_W = {"o": "self_attn.o_proj.weight", "down": "mlp.down_proj.weight"}

    def lm_w(self, i, kind):
        if self.fp8 is not None:
            return self.fp8.weight(i, kind)
        if kind == "qkv":
            return self.store.qkv(i)
        if kind == "gate_up":
            return self.store.gate_up(i)
        return self.store.p(f"lang_model.model.layers.{i}.{self._LM_W[kind]}")

    def lm_linear(self, x, i, kind, out=None, R=None, epilogue=ops.EPI_STORE):
        """y = x W^T (+ R) with W = decoder layer i's `kind` operand (forward only)"""
        if self.fp8 is not None:
            return self.fp8.linear(x, i, kind, out=out, R=R, epilogue=epilogue)
        return ops.gemm_bf16(ops.NT, x, self.lm_w(i, kind), out=out, R=R, epilogue=epilogue)

    @torch.no_grad()
    def to_fp8_weight_only(self, resident_bf16=False, gemm_mode=None):
        """Deployment form for inference: decoder Linear weights -> e4m3fn codes + per-output-channel
        scales (navillm_amd/fp8.py); the bf16 copies of those weights and ALL gradient buffers are released.  Irreversible;
        training / backward / state_dict() of the decoder layers are not available afterwards.
        resident_bf16=True trades the memory back for time: the bf16 buffers are kept, overwritten with the de-quantised operand
        bf16(s*q) -- the SAME values the per-call de-quantisation produces -- so prefill / K/V-reuse GEMMs skip the pre-pass while the
        decode steps still stream the codes (13B: 12.7 GB codes + 25.4 GB bf16 of 288 GB).
        gemm_mode (7 | 9, default NAVILLM_FP8_GEMM_MODE or 7): see fp8.Fp8DecoderWeights -- how the few-hundred-row GEMMs of K/V-reuse
        steps consume the codes (nv_gemm_fp8w)."""
        from .fp8 import Fp8DecoderWeights
        if self.fp8 is not None:
            return self.fp8
        torch.cuda.synchronize(self.device)
        self.fp8 = Fp8DecoderWeights(self, resident_bf16=resident_bf16, gemm_mode=gemm_mode)
        if resident_bf16:
            self.store.release_grads(self._named)
            self.eval()
            torch.cuda.empty_cache()
            return self.fp8
        self.store.release_decoder_layers_and_grads(self._named)
        self.eval()
        torch.cuda.empty_cache()
        return self.fp8

    @torch.no_grad()
    def fp8_release_resident(self):
        """`to_fp8_weight_only(resident_bf16=True)` -> the memory-lean form: drop the resident bf16(s*q) operands (and the compacted
        store that goes with it); prefill GEMMs de-quantise into the shared scratch panel from then on."""
        f8 = self.fp8
        if f8 is None or f8.resident is None:
            return f8
        torch.cuda.synchronize(self.device)
        f8.resident = None
        n_max = max(q.shape[0] * q.shape[1] for q in f8.codes[0].values())
        f8._scratch = torch.empty((n_max,), dtype=BF16, device=self.device)
        f8.overlap = os.environ.get("NAVILLM_FP8_OVERLAP", "0") == "1"
        self.store.release_decoder_layers_and_grads(self._named)
        if self.kv is not None:
            self.kv._dec_key = None                                # the native decoder's weight table points at the released buffers
        torch.cuda.empty_cache()
        return f8

    def reserve_activations(self, batch, seq_len):
        """size the LM activation arena once, up front (B*S rows)"""
        self.arena.reserve(batch * seq_len)

    def enable_kv_cache(self, batch_size, capacity=1024):
        """Inference: keep every decoder layer's K/V of the prompts seen last.  Under
        `torch.no_grad()` the navigation / object-grounding modes then run only the tokens that differ from the previous
        call of the same batch slot (the prompt prefix up to the last <hist> repeats from step to step).  Call
        `reset_kv_cache()` at episode boundaries (a mismatching prompt is detected anyway and simply recomputed)."""
        from .kvcache import KVCacheLM
        self.kv = KVCacheLM(self, batch_size, capacity)
        return self.kv

    # ---- optional training mode: the prompt's static prefix is computed once per episode (navillm_amd/episode.py)
    def begin_episode(self, prefix_ids, capacity=1024, teacher_forced=False, max_length=None, accumulate=1):
        """prefix_ids: B lists of token ids -- the part of every navigation prompt of the coming episode that never changes
        (everything up to "### History:").  Until `finish_episode()`, training-mode `model('navigation')` calls push only the rest
        of each prompt through the LM, over the cached prefix (every other mode -- object_grounding included -- takes the full `_lm`
        path: its gradients land in `.grad` at once, as always); `finish_episode()` runs the prefix's one
        backward -- and, in the default `NAVILLM_EPISODE_DEFER=all` form, the steps' LM backward with it: ALL parameter gradients of
        the episode appear there, the steps' `backward()` calls only record their output gradients (navillm_amd/episode.py).
        Exact up to bf16 rounding order; call it before `optimizer.step()`.
        teacher_forced=True (round 4; imitation-learning rollouts only, mp3d_agent.py:514 `feedback="teacher"`): the steps' LM FORWARD is
        deferred as well -- `model('navigation')` returns `fuse_embeds` at once (the next step's history token) and a
        `losses.DeferredLogits` handle as `fuse_logits`; `criterion(handle, targets) * coef / B` and `.backward()` work on it as on a
        tensor (they record targets and scale), and `finish_episode()` pushes ALL the steps through the decoder as one batch before
        the batched backward.  The logits / loss values exist afterwards (`handle.value`, `float(loss)`).
        accumulate=n > 1 (round 5; teacher-forced episodes only) = the training loop's `gradient_accumulation_step` (train.py:68,86-89;
        scripts/multi_wo_pretrain.sh:16 runs `--batch_size 1 --gradient_accumulation_step 8`): the optimizer steps only after every
        n-th episode, so the n episodes of a window are batched with one another as the steps of one episode are --
        `finish_episode()` of the first n - 1 returns at once, the n-th (or whatever needs `.grad` first: `FlatAdamW.clip_grad_norm_` /
        `step`, `model.parameters()`, an episode opened without `accumulate`) runs ONE forward + backward over the rows of all n
        (navillm_amd/episode.py::_begin_window).  Logits / loss values exist after that."""
        from .episode import PrefixEpisode
        if self._auto_open:            # an explicit begin_episode() takes precedence: the automatic episode hands its gradients over first
            self._auto_close("begin_episode")
        B = len(prefix_ids)
        if max_length is None:         # the tokenizer's left-truncation limit (modified_lm.py:57); the stub tokenizer of the synthetic driver: 1024
            max_length = int(getattr(self.lang_model, "max_length", 1024))
        accumulate = int(accumulate or 1)
        if accumulate > 1 and teacher_forced and os.environ.get("NAVILLM_EPISODE_DEFER", "all") == "all" and \
                os.environ.get("NAVILLM_EPISODE_WINDOW", "1") != "0":
            w = self._window
            if w is not None and (w.Bs != B or w.B != B * accumulate or w.cap != capacity):
                w.flush_window()
                w = None
            if self.episode is not None and self.episode is not w:
                self.episode.assert_no_pending_gradients("begin_episode()")
                if self.episode is not self._window:
                    self._episode_plain = self.episode
            if w is None:
                w = PrefixEpisode(self, B * accumulate, capacity, max_length, samples_per_episode=B)
            self._window = self.episode = w
            w.max_length = int(max_length)
            w.begin(prefix_ids, teacher_forced=True, window=True)
            return w
        if self._window is not None:
            self._window.flush_window()                # an episode outside the window: the window's gradients go to .grad first
            if self.episode is self._window:
                self.episode = self._episode_plain
        if self.episode is not None:
            # a begin before the previous episode's finish would silently drop that episode's deferred gradients
            self.episode.assert_no_pending_gradients("begin_episode()")
        if self.episode is None or self.episode.B != B or self.episode.cap != capacity:
            self.episode = PrefixEpisode(self, B, capacity, max_length)
        self.episode.max_length = int(max_length)
        self.episode.begin(prefix_ids, teacher_forced=teacher_forced)
        return self.episode

    def finish_episode(self):
        self._auto_open = False
        if self.episode is not None:
            self.episode.finish()

    # ---- automatic episodes (round 6): the fast path behind the reference's OWN call pattern
    # The rollout of tasks/agents/mp3d_agent.py:660-778 knows nothing of episodes: per step `model('navigation', ...)`, `criterion`,
    # `backward()`; train.py:86-89 then clips through `model.parameters()` and steps.  So the model opens the episode itself on the
    # first grad-enabled training-mode navigation call (prefix = the static part of each incoming prompt; `fuse_logits` is a
    # losses.LazyLogits handle on which `torch.softmax(nav_logits / T, 1)` at mp3d_agent.py:732 stays lazy and which runs the pending
    # steps the moment sampled / argmax feedback reads it -- or, NAVILLM_AUTO_EPISODE=step, a plain tensor with every forward at once),
    # later calls whose prompts start with the same prefixes extend it, and it is closed -- its deferred backward run, every gradient
    # handed to `.grad` -- by whatever comes first: a navigation call with OTHER prompts (the next rollout), `model.parameters()` /
    # `named_parameters()` (torch.nn.utils.clip_grad_norm_, train.py:87; also through the NavDataParallel wrapper),
    # `FlatAdamW.clip_grad_norm_` / `step`, an explicit `begin_episode()`, `state_dict()`-free code never needs it.
    # `zero_grad()` while such an episode still HOLDS gradients raises (they would leak into the next optimizer step).
    # NAVILLM_AUTO_EPISODE=0 switches it off (the reference's formulation, every step through the full prompt).
    def _episode_active(self):
        """is there an open episode the next grad-enabled navigation step belongs to?  (An accumulation window whose episodes are all
        finished is open but has no CURRENT episode: a step without begin_episode() then takes the automatic / full path -- ADVICE r5.)"""
        ep = self.episode
        if ep is None or ep.prefix is None:
            return False
        P = ep.prefix
        return not (P.get("window") and P["finished"] >= P["episodes"])

    def _auto_prefix_ids(self, batch, ids_l):
        """the static prefix of every incoming prompt, or None when it cannot be told (then the step takes the full `_lm` path):
        `batch['prefix_lens']` (callers that tokenise themselves, e.g. the synthetic driver) or the prompt STRINGS + the attached
        tokenizer (the reference's agents: everything up to "### History:", navillm_amd/prompts.py::static_prefix)"""
        cfg = self.cfg
        pl = batch.get("prefix_lens")
        if pl is not None:
            pre = [list(ids_l[b][:int(pl[b])]) for b in range(len(ids_l))]
        elif batch.get("prompts") is not None and self.lang_model.tokenizer is not None and batch.get("input_ids") is None:
            from .episode import prefix_ids_from_prompts
            try:
                pre = prefix_ids_from_prompts(self.lang_model.tokenizer, list(batch["prompts"]))
            except ValueError:             # a prompt without the "### History:" marker
                return None
            pre = [p if ids_l[b][:len(p)] == p else None for b, p in enumerate(pre)]     # (a left-truncated prompt no longer starts with it)
            if any(p is None for p in pre):
                return None
        else:
            return None
        special = set(cfg.special_token_ids)
        limit = int(getattr(self.lang_model, "max_length", 1024))
        for b, p in enumerate(pre):
            if not (0 < len(p) < min(len(ids_l[b]), limit, 1024)) or len(ids_l[b]) >= limit or any(t in special for t in p):
                return None
        return pre

    def _auto_matches(self, ids_l):
        """do these prompts belong to the open automatic episode?  (same batch size, every prompt starts with its registered prefix --
        or was left-truncated at the tokenizer's limit, which `PrefixEpisode.fits` then sends through the full path)"""
        ep = self.episode
        P = None if ep is None else ep.prefix
        if P is None or P.get("window") or len(ids_l) != ep.Bs:
            return False
        for j, ids in enumerate(ids_l):
            lp = int(P["lens"][j])
            if not (ids[:lp] == P["ids"][j] and lp < len(ids)) and len(ids) < min(ep.cap, ep.max_length):
                return False
        return True

    def _auto_close(self, why="caller"):
        """hand the automatic episode's gradients over to `.grad` (its one deferred backward); a no-op without one"""
        if not self._auto_open:
            return
        self._auto_open = False
        self.auto_stats["closed_by"][why] = self.auto_stats["closed_by"].get(why, 0) + 1
        if self.episode is not None and self.episode.prefix is not None:
            self.episode.finish()

    def grad_handover(self, why="caller"):
        """everything that is still deferred becomes `.grad` NOW: the open accumulation window and the automatic episode.  Called by
        whoever is about to read the gradients (FlatAdamW, `parameters()`, NavDataParallel.parameters())."""
        w = getattr(self, "_window", None)
        todo_w = w is not None and w.window_open()
        todo_a = getattr(self, "_auto_open", False)
        if not (todo_w or todo_a):
            return
        # whoever is about to clip / step is past the optimizer step's LAST backward: under data parallelism the deferred work that runs
        # now IS that backward, so it runs inside final_backward() and its per-layer exchange is launched from the deferred walk, overlapped
        # -- an unmodified training loop gets the overlap without knowing the context exists.  (A handover that turns out NOT to be the last
        # backward -- model.parameters() called for logging -- costs a second exchange at the optimizer, not correctness: the mean over
        # ranks is linear, and every rank then holds mean(A) + its later local gradients.)
        dp = getattr(self, "_dp", None)
        ctx = dp.final_backward if (dp is not None and why in ("optimizer", "parameters") and hasattr(dp, "final_backward")) else contextlib.nullcontext
        with ctx():
            if todo_w:
                w.flush_window()
            if todo_a:
                self._auto_close(why)

    def flush_accumulation_window(self):
        """hand over the gradients of the open accumulation window (`begin_episode(..., accumulate=n)`) now; a no-op without one.
        Called by FlatAdamW before the clip / the update and by `parameters()`."""
        if self._window is not None:
            self._window.flush_window()

    def episode_release(self):
        """give the prefix-reuse episode buffers back (tens of GB after long episodes); the next begin_episode() re-creates them"""
        self.grad_handover("episode_release")
        for ep in {id(e): e for e in (self.episode, self._window, self._episode_plain) if e is not None}.values():
            ep.assert_no_pending_gradients("episode_release()")
            ep.prefix = None
            ep._cursor = 0
            ep.release_buffers()

    def episode_abort(self):
        """drop an episode that was begun but will not be finished.  Its deferred gradients are discarded -- but whatever the
        episode has ALREADY written into `.grad` is not undone: flushed segments of a long episode (`flush_segment`: LM weight,
        embedding and encoder gradients of those steps), steps that ran their backward at once (the non-deferred forms, truncated
        prompts), while the prefix's own backward never runs.  `.grad` then holds a partial, inconsistent sum, so the gradient
        store is marked tainted: `FlatAdamW.clip_grad_norm_` / `step` raise until `zero_grad()`."""
        self._auto_open = False
        ep = self.episode
        if ep is not None:
            P = ep.prefix
            if P is not None and (P.get("segments", 0) > 0 or P.get("kv_steps", 0) > 0 or ep.stats.get("recomputed_steps", 0) > 0):
                self.store.tainted = "episode_abort() after part of the episode's gradients had been written to .grad"
            if P is not None and P.get("window") and P.get("finished", 0) > 0:
                # the FINISHED episodes of an accumulation window hold gradients that exist nowhere else: dropping them silently would
                # make the optimizer step on a partial sum
                self.store.tainted = "episode_abort() dropped the deferred gradients of the accumulation window's finished episodes"
            ep.prefix = None
            ep._cursor = 0

    def named_parameters(self, *args, **kwargs):
        """nn.Module.named_parameters (and, through it, `parameters()`), except that whoever asks for the parameters may be about to read
        `.grad`: the only caller inside the reference's training loop is `torch.nn.utils.clip_grad_norm_(model.parameters(), 40.)`
        (train.py:87).  So an open accumulation window and an AUTOMATIC episode hand their gradients over first; an episode the
        caller opened explicitly and that still HOLDS gradients (its steps ran backward(), finish_episode() has not run) raises --
        the clip would see, and the optimizer then step on, the encoder's gradients alone.  (FlatAdamW guards its own clip / step the
        same way.)"""
        self.grad_handover("parameters")
        if getattr(self, "episode", None) is not None:
            self.episode.assert_no_pending_gradients("model.parameters() [e.g. torch.nn.utils.clip_grad_norm_(model.parameters(), ...)]")
        if getattr(self, "store", None) is not None:
            self.store.wait_params()           # an optimizer update still on its side stream: the caller is about to read parameters / .grad
        return super().named_parameters(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        self.store.wait_params()
        return super().state_dict(*args, **kwargs)

    def _lm_episode(self, ids_cpu, am_cpu, cand_vis=None, hist_vis=None, obj_vis=None, layout=None):
        """-> [B, d] over the cached prefix, or None when a prompt of this step was left-truncated at the tokenizer's max_length
        (modified_lm.py:77-87): the caller then runs the step through the full `_lm` path (PrefixEpisode.fits)"""
        ids_l, vix_l, vis_all, _ = layout if layout is not None else self._vis_layout(ids_cpu, am_cpu, cand_vis, hist_vis, obj_vis)
        if not self.episode.fits(ids_l):
            return None
        return self.episode.lm(ids_l, vix_l, vis_all)

    def reset_kv_cache(self):
        if self.kv is not None:
            self.kv.reset()

    def zero_grad(self, set_to_none=False):
        self.assert_no_auto_pending("zero_grad()")
        self.store.zero_grad()

    def assert_no_auto_pending(self, what):
        """an AUTOMATIC episode that still holds gradients at zero_grad(): nothing handed them to `.grad` (no clip through
        `model.parameters()`, no FlatAdamW) -- zeroing now would make them part of the NEXT optimizer step"""
        if self._auto_open and self.episode is not None and self.episode.has_pending_gradients():
            raise RuntimeError(f"{what} while the automatic prefix-reuse episode still holds this step's LM gradients: nothing read "
                               "them (torch.nn.utils.clip_grad_norm_(model.parameters(), ...) or FlatAdamW.clip_grad_norm_ / step hand "
                               "them over; with another optimizer call model.grad_handover() before optimizer.step(), or set "
                               "NAVILLM_AUTO_EPISODE=0)")

    def _apply(self, fn, recurse=True):
        # .to(device)/.cuda() on the right device are no-ops; anything else would break the flat views
        probe = fn(torch.empty(0, device=self.device))
        if probe.device != self.device or probe.dtype != torch.float32:
            raise RuntimeError("NavModel lives in flat HBM buffers: .to(other device/dtype) is not supported")
        return self

    # ---- DP hooks (navillm_amd/parallel.py installs itself here)
    def _dp_begin_backward(self):
        if self._dp is not None:
            self._dp.on_backward_begin()

    def _dp_layer_done(self, i, events=()):
        if self._dp is not None:
            self._dp.on_layer_done(i, events)

    def wgrad_stream(self):
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device)
        return self._wgrad_stream

    # ------------------------------------------------------------------ small helpers
    def P(self, name):
        return self._named[name]

    def _Prow(self, name, i):
        """row i of an embedding table used as a plain torch index (autograd accumulates into the flat .grad view itself)"""
        if torch.is_grad_enabled():
            self.store.touch(name)
        return self._named[name][i]

    def _lin(self, x, prefix):
        return Fn.linear(x, self.P(prefix + ".weight"), self.P(prefix + ".bias"))

    def _ln(self, x, prefix, eps):
        return Fn.layer_norm(x, self.P(prefix + ".weight"), self.P(prefix + ".bias"), eps)

    def _seq2(self, x, prefix):
        return self._ln(self._lin(x, prefix + ".0"), prefix + ".1", 1e-12)

    def _drop(self, x, p, key):
        m = None if self.injected_dropout is None else self.injected_dropout.get(key)
        return Fn.dropout(x, p, self.training, m)

    def _i32(self, t):
        return ops.h2d(t, self.device, torch.int32).contiguous()

    # ------------------------------------------------------------------ forward dispatch (nav_model.py:96-126)
    def forward(self, mode, batch, **kwargs):
        batch = collections.defaultdict(lambda: None, batch)
        if mode == "panorama":
            v = batch["view_img_fts"]
            if self.training and self.drop_env_p > 0:
                v = self._drop(v, self.drop_env_p, "drop_env.view")
            o = batch["obj_img_fts"]
            if o is not None and self.training and self.drop_env_p > 0:
                o = self._drop(o, self.drop_env_p, "drop_env.obj")
            return self.forward_panorama_per_step(v, batch["view_lens"], batch["loc_fts"], batch["nav_types"], o,
                                                  batch["obj_lens"], batch["obj_loc_fts"])
        elif mode == "navigation":
            return self.forward_navigation(mode, batch, **kwargs)
        elif mode == "summarization" or mode == "embodied_qa":
            return self.forward_summarization(mode, batch, **kwargs)
        elif mode == "3dqa":
            return self.forward_3dqa(mode, batch, **kwargs)
        elif mode == "object_grounding":
            return self.forward_object_grounding(mode, batch, **kwargs)
        else:
            raise NotImplementedError("wrong mode: %s" % mode)

    # ------------------------------------------------------------------ scene encoder (image_embedding.py:51-121)
    def forward_panorama_per_step(self, view_img_fts, view_lens, loc_fts=None, nav_types=None, obj_img_fts=None,
                                  obj_lens=None, obj_loc_fts=None):
        cfg = self.cfg
        e = "img_embeddings"
        view_img_fts = ops.h2d(view_img_fts, self.device, F32)
        B, N, _ = view_img_fts.shape
        h = cfg.enc_hidden_size
        x = self._ln(self._lin(view_img_fts, e + ".img_linear"), e + ".img_layer_norm", 1e-12)
        if loc_fts is None:
            loc_fts = torch.zeros((B, N, 7), dtype=F32, device=self.device)
        x = Fn.add(x, self._ln(self._lin(ops.h2d(loc_fts, self.device, F32), e + ".loc_linear"), e + ".loc_layer_norm", 1e-12))
        if nav_types is None:
            nav_types = torch.ones((B, N), dtype=torch.int32, device=self.device)
        x = Fn.EmbedAddF32.apply(self.P(e + ".nav_type_embedding.weight"), self._i32(nav_types).view(-1), x)
        x = self._ln(x, e + ".layer_norm", 1e-12)
        x = self._drop(x, cfg.enc_dropout, "emb.drop")
        lens_dev = ops.h2d(view_lens, self.device)
        pano_masks = torch.arange(N, device=self.device).unsqueeze(0) < lens_dev.unsqueeze(1)
        if cfg.num_pano_layers > 0:
            if cfg.fuse_obj:
                # object tokens join the encoder sequence (image_embedding.py:78-94): per sample
                # [views[:len_v] ; objects[:len_o]], padded to the batch max, view part sliced back afterwards.
                # The ragged concat / slice-back are host-built row tables (lens are host data in the agent).
                vl = torch.as_tensor(view_lens).cpu().long()
                olh = torch.as_tensor(obj_lens).cpu().long()
                O = obj_img_fts.shape[1]
                o = self._seq2(ops.h2d(obj_img_fts, self.device, F32), e + ".obj_linear")
                o = Fn.add(o, self._ln(self._lin(ops.h2d(obj_loc_fts, self.device, F32), e + ".loc_linear"),
                                       e + ".loc_layer_norm", 1e-12))
                o = Fn.add(o.view(B * O, h), self._Prow(e + ".nav_type_embedding.weight", 2))
                T_ = int((vl + olh).max())
                src_v = torch.full((B * T_,), -1, dtype=torch.int32)
                src_o = torch.full((B * T_,), -1, dtype=torch.int32)
                inv_v = torch.full((B * N,), -1, dtype=torch.int32)
                inv_o = torch.full((B * O,), -1, dtype=torch.int32)
                back = torch.full((B * N,), -1, dtype=torch.int32)
                back_inv = torch.full((B * T_,), -1, dtype=torch.int32)
                for b in range(B):
                    for j in range(int(vl[b])):
                        src_v[b * T_ + j] = b * N + j
                        inv_v[b * N + j] = b * T_ + j
                        back[b * N + j] = b * T_ + j
                        back_inv[b * T_ + j] = b * N + j
                    for j in range(int(olh[b])):
                        src_o[b * T_ + int(vl[b]) + j] = b * O + j
                        inv_o[b * O + j] = b * T_ + int(vl[b]) + j
                fuse = Fn.GatherRowsF32.apply(x.view(B * N, h), ops.h2d(src_v, self.device), ops.h2d(inv_v, self.device), None)
                fuse = Fn.GatherRowsF32.apply(o, ops.h2d(src_o, self.device), ops.h2d(inv_o, self.device), fuse)
                fuse = self._pano_encoder(fuse.view(B, T_, h), ops.h2d(vl + olh, self.device, torch.int32))
                x = Fn.GatherRowsF32.apply(fuse.view(B * T_, h), ops.h2d(back, self.device), ops.h2d(back_inv, self.device),
                                           None).view(B, N, h)
            else:
                x = self._pano_encoder(x, lens_dev.to(torch.int32).contiguous())
        x = self._lin(x, e + ".mapper")
        x = Fn.RowScaleF32.apply(x, pano_masks.to(F32).view(-1).contiguous()).view(B, N, cfg.hidden_size)
        ret = {"pano_embeds": x, "pano_masks": pano_masks}
        if obj_img_fts is not None and obj_img_fts.shape[1] > 0:
            oe = self._seq2(ops.h2d(obj_img_fts, self.device, F32), e + ".obj_projector")
            ol = ops.h2d(obj_lens, self.device)
            obj_masks = torch.arange(obj_img_fts.shape[1], device=self.device).unsqueeze(0) < ol.unsqueeze(1)
            assert oe.shape[:2] == obj_loc_fts.shape[:2], \
                f"shape of obj_embeds {oe.shape[:2]} must equal to shape of obj_loc_fts {obj_loc_fts.shape[:2]}"
            ret.update({"obj_embeds": oe, "obj_loc_fts": obj_loc_fts, "obj_masks": obj_masks})
        return ret

    def _pano_encoder(self, x, lens_i32):
        """pre-norm TransformerEncoder + final LayerNorm(1e-12) (detr_transformer.py:71-89,170-182; ops.py:6-18)"""
        cfg = self.cfg
        e = "img_embeddings"
        B, N, h = x.shape
        heads, hd = cfg.enc_num_heads, h // cfg.enc_num_heads
        for i in range(cfg.num_pano_layers):
            p = f"{e}.pano_encoder.layers.{i}"
            y = self._ln(x, p + ".norm1", 1e-5)
            qkv = Fn.linear(y, self.P(p + ".self_attn.in_proj_weight"), self.P(p + ".self_attn.in_proj_bias"))
            am = None if self.injected_dropout is None else self.injected_dropout.get(f"l{i}.attn")
            a = Fn.mha(qkv.view(B * N, 3 * h), lens_i32, B, N, heads, hd, cfg.enc_dropout, self.training, am).view(B, N, h)
            a = self._lin(a, p + ".self_attn.out_proj")
            x = Fn.add(x, self._drop(a, cfg.enc_dropout, f"l{i}.drop1"))
            y = self._ln(x, p + ".norm2", 1e-5)
            y = Fn.GeluF32.apply(self._lin(y, p + ".linear1"))
            y = self._drop(y, cfg.enc_dropout, f"l{i}.drop")
            y = self._lin(y, p + ".linear2")
            x = Fn.add(x, self._drop(y, cfg.enc_dropout, f"l{i}.drop2"))
        return self._ln(x, e + ".pano_encoder.norm", 1e-12)

    # ------------------------------------------------------------------ the visual-token LM (modified_lm.py:89-146)
    def _tokens(self, batch, text):
        """ids/mask on the host: pre-tokenised (synthetic driver) or via the attached tokenizer."""
        if batch.get("input_ids") is not None:
            ids, am = torch.as_tensor(batch["input_ids"]).cpu(), torch.as_tensor(batch["attention_mask"]).cpu()
            tt = batch.get("token_type_ids")
            return ids, am, (None if tt is None else torch.as_tensor(tt).cpu())
        tok = self.lang_model.tokenize(text)
        return tok["input_ids"], tok["attention_mask"], tok.get("token_type_ids")

    def _lm(self, ids_cpu, am_cpu, cand_vis=None, hist_vis=None, obj_vis=None, cls_tail=False):
        """-> hidden states (post final RMSNorm), bf16.  Rows: with `pack_rows` (default) only the REAL tokens of the batch,
        sample after sample (`self._row_map` = their indices in the reference's flattened [B*S] layout; the left-padding
        rows are never computed); otherwise all B*S rows.  With cls_tail and every <cls_1> at the last position (always
        true for the left-padded navigation / grounding prompts): only those B rows, [B, d]."""
        cfg = self.cfg
        B, S = ids_cpu.shape
        # host-side index bookkeeping in numpy: torch's CPU ops on these few-thousand-element tensors go through the
        # intra-op thread pool, which on a 256-thread host now and then takes 80-100 ms to get through a barrier
        # (tools/pack_probe.py) -- more than the whole LM forward
        am_np = am_cpu.numpy().astype(bool)
        lens_np = am_np.sum(1)
        kv_np = (S - lens_np).astype(np.int32)
        assert bool((am_np == (np.arange(S)[None] >= kv_np[:, None])).all()), "attention_mask must be left padding"
        kv_start = torch.from_numpy(kv_np)
        if self.flop_log is not None:
            self.flop_log.append(("lm", int(lens_np.sum()), int((lens_np.astype(np.int64) ** 2).sum()), torch.is_grad_enabled()))
        packed = None
        self._row_map = None
        flat = ids_cpu.reshape(-1)
        if self.pack_rows:
            keep_np = np.flatnonzero(am_np.reshape(-1))
            cu_np = np.zeros(B + 1, dtype=np.int32)
            cu_np[1:] = np.cumsum(lens_np)
            pos_np = np.broadcast_to(np.arange(S, dtype=np.int32)[None], (B, S)).reshape(-1)[keep_np]
            if self.rope_frame == "sample":
                # parity instrument (tests): positions counted from each sample's first token, the frame of the prefix-reuse /
                # K/V-cache paths, instead of the reference's arange(S) over the left padding
                pos_np = pos_np - np.repeat(kv_np, lens_np)
                kv_start = torch.zeros(B, dtype=torch.int32)
            keep = torch.from_numpy(keep_np)
            flat = torch.from_numpy(ids_cpu.numpy().reshape(-1)[keep_np])
            self._row_map = keep
            packed = (ops.h2d(torch.from_numpy(cu_np), self.device), ops.h2d(torch.from_numpy(np.ascontiguousarray(pos_np)), self.device),
                      int(lens_np.max()))
            last_rows = torch.from_numpy((cu_np[1:] - 1).astype(np.int32))
        else:
            last_rows = torch.arange(B, dtype=torch.int32) * S + (S - 1)
        M = flat.numel()
        flat_np = flat.numpy()
        vis_idx_np = np.full((M,), -1, dtype=np.int32)
        parts, rows, off = [], [], 0
        for tok_id, vis in ((cfg.cand_token_id, cand_vis), (cfg.hist_token_id, hist_vis), (cfg.obj_token_id, obj_vis)):
            loc = np.flatnonzero(flat_np == tok_id)
            if loc.size == 0:
                continue
            assert vis is not None and vis.shape[0] == loc.size, \
                f"{loc.size} special tokens of id {tok_id} but {None if vis is None else vis.shape[0]} visual rows"
            vis_idx_np[loc] = np.arange(off, off + loc.size, dtype=np.int32)
            rows.append(loc.astype(np.int32))
            parts.append(vis.to(F32))
            off += loc.size
        vis_idx = torch.from_numpy(vis_idx_np)
        rows = [torch.from_numpy(np.concatenate(rows))] if rows else []
        vis_all = torch.cat(parts, 0).contiguous() if parts else None
        vis_rows = ops.h2d(torch.cat(rows), self.device) if rows else None
        E = Fn.EmbedVis.apply(vis_all, self._anchor if torch.is_grad_enabled() else None, self,
                              ops.h2d(torch.from_numpy(flat_np.astype(np.int32)), self.device), ops.h2d(vis_idx, self.device), vis_rows, flat)
        tail = None
        ids_np = ids_cpu.numpy()
        if cls_tail and self.prune_last_layer and bool((ids_np[:, -1] == cfg.cls_token_ids[0]).all()) \
                and int((ids_np == cfg.cls_token_ids[0]).sum()) == B:
            tail = ops.h2d(last_rows, self.device)
        Hs = Fn.LlamaStack.apply(E, self, B, S, ops.h2d(kv_start, self.device), tail, packed)
        if cls_tail and tail is None:
            Hs = Fn.GatherRowsBF16.apply(Hs, self._cls_rows(flat))
        return Hs

    def _vis_layout(self, ids_cpu, am_cpu, cand_vis, hist_vis, obj_vis, hist_keys=None):
        """per-sample unpadded ids + visual-row indices (rows of cat(cand, hist, obj), the order of modified_lm.py:100-110)
        + one reuse key per visual row (`False`: recompute every call)."""
        cfg = self.cfg
        B, S = ids_cpu.shape
        flat = ids_cpu.reshape(-1)
        vis_idx = torch.full((B * S,), -1, dtype=torch.int64)
        parts, keys, off = [], [], 0
        for tok_id, vis, k in ((cfg.cand_token_id, cand_vis, None), (cfg.hist_token_id, hist_vis, hist_keys),
                               (cfg.obj_token_id, obj_vis, None)):
            loc = torch.nonzero(flat == tok_id).view(-1)
            if loc.numel() == 0:
                continue
            assert vis is not None and vis.shape[0] == loc.numel(), \
                f"{loc.numel()} special tokens of id {tok_id} but {None if vis is None else vis.shape[0]} visual rows"
            vis_idx[loc] = torch.arange(off, off + loc.numel())
            parts.append(vis.to(F32))
            keys.extend(k if k is not None else [False] * loc.numel())
            off += loc.numel()
        vis_all = torch.cat(parts, 0).contiguous() if parts else None
        lens = am_cpu.bool().sum(1).tolist()
        ids_l = [ids_cpu[b, S - lens[b]:].tolist() for b in range(B)]
        vix_l = [vis_idx.view(B, S)[b, S - lens[b]:].tolist() for b in range(B)]
        return ids_l, vix_l, vis_all, keys

    def _lm_cached(self, ids_cpu, am_cpu, cand_vis=None, hist_vis=None, obj_vis=None, hist_keys=None):
        """no-grad twin of `_lm(..., cls_tail=True)` over the K/V cache: [B, d] hidden state of each prompt's last token."""
        assert bool((ids_cpu[:, -1] == self.cfg.cls_token_ids[0]).all()), "cached path: prompts must end in <cls_1>"
        ids_l, vix_l, vis_all, keys = self._vis_layout(ids_cpu, am_cpu, cand_vis, hist_vis, obj_vis, hist_keys)
        return self.kv.extend(ids_l, vix_l, vis_all, keys)

    @torch.no_grad()
    def _generate(self, ids_cpu, am_cpu, cand_vis, hist_vis, max_new_tokens=20, trie=None, do_sample=False, temperature=1.0,
                  top_k=50, **unused):
        """`self.lang_model.generate(...)` of nav_model.py:324-341,388-402: eos-terminated, pad = unk, special ids masked,
        optional trie; greedy, or HF's sampling when the caller forwards `do_sample` / `temperature` (llava.py:58-62); served by
        the K/V-cache decoder (navillm_amd/kvcache.py)."""
        from .kvcache import KVCacheLM
        B, S = ids_cpu.shape
        need = S + max_new_tokens
        # reuse the cache object of the previous call when it fits: its buffers, its native layer table and the captured decode
        # graph (evaluation calls generate() once per batch with the same B and max_new_tokens)
        kv = self.kv if (self.kv is not None and self.kv.B == B and self.kv.cap >= need) else self._gen_kv
        if kv is None or kv.B != B or kv.cap < need:
            kv = self._gen_kv = KVCacheLM(self, B, capacity=(need + 127) // 128 * 128)
        ids_l, vix_l, vis_all, _ = self._vis_layout(ids_cpu, am_cpu, cand_vis, hist_vis, None)
        tok = self.lang_model.tokenizer
        eos = getattr(tok, "eos_token_id", None)
        pad = getattr(tok, "unk_token_id", None)
        gen = kv.generate(ids_l, vix_l, vis_all, max_new_tokens=max_new_tokens, eos_token_id=2 if eos is None else eos,
                          pad_token_id=0 if pad is None else pad, trie=trie, do_sample=do_sample, temperature=temperature, top_k=top_k)
        kv.reset()
        out = {"generated_ids": gen}
        if hasattr(tok, "batch_decode"):
            out["generated_sentences"] = tok.batch_decode(gen, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        else:
            out["generated_sentences"] = [" ".join(str(t) for t in g) for g in gen]
        return out

    def _lm_loss(self, Hs, ids_cpu, labels_cpu):
        """shifted mean CE over labels != -100 (modified_lm.py:126-137)."""
        B, S = ids_cpu.shape
        shift = torch.full((B, S), -100, dtype=torch.int64)
        shift[:, :-1] = labels_cpu[:, 1:]
        n_valid = int((shift != -100).sum())
        shift = shift.view(-1)
        if self.flop_log is not None:
            self.flop_log.append(("lm_head", int(Hs.shape[0]), 0, torch.is_grad_enabled()))
        if self._row_map is not None:          # Hs holds the real tokens only (see _lm); a padding row never carries a label
            assert int((shift[self._row_map] != -100).sum()) == n_valid
            shift = shift[self._row_map]
        return Fn.LMHeadLoss.apply(Hs, self, ops.h2d(shift.contiguous(), self.device, torch.int32), n_valid)

    @staticmethod
    def _stack_hist(hist_vis):
        flat = [v for vis in hist_vis for v in vis]
        return torch.stack(flat, 0) if flat else None

    _hist_serial = itertools.count(1)

    @classmethod
    def _hist_keys(cls, hist_vis):
        """one reuse key per history row.  History embeddings are constants once appended (mp3d_agent.py:774-778) and the
        agent passes the SAME tensor objects step after step, so each object gets a serial number the first time it is seen.
        (Not `data_ptr()`: after an episode ends the caching allocator hands the same address to a different tensor, and a
        repeated instruction would then silently reuse stale K/V -- ADVICE r1.)  A caller that rebuilds its history tensors
        every step just gets fresh serials, i.e. a recompute."""
        keys = []
        for b, vis in enumerate(hist_vis):
            for k, v in enumerate(vis):
                sid = getattr(v, "_nv_hist_id", None)
                if sid is None:
                    sid = next(cls._hist_serial)
                    v._nv_hist_id = sid
                keys.append(("hist", b, k, sid))
        return keys

    def _cls_rows(self, flat_ids_cpu):
        loc = torch.nonzero(flat_ids_cpu.reshape(-1) == self.cfg.cls_token_ids[0]).view(-1)
        return ops.h2d(loc, self.device, torch.int32)

    # ------------------------------------------------------------------ navigation (nav_model.py:129-247)
    def forward_navigation(self, mode, batch, training=True, **kwargs):
        cfg, dev = self.cfg, self.device
        d = cfg.hidden_size
        vp_img = batch["vp_img_embeds"]
        B, Nv, _ = vp_img.shape
        g_img, g_step, g_pos = batch["gmap_img_embeds"], batch["gmap_step_ids"], batch["gmap_pos_fts"]
        G = g_img.shape[1]
        # host copies of the two bool masks drive the python-side matching, as in the reference; a caller
        # that already has them on the host can pass them along and skip the D2H sync
        gm_cpu = batch["_gmask_cpu"] if batch["_gmask_cpu"] is not None else torch.as_tensor(batch["gmap_masks"]).cpu().bool()
        gv_cpu = batch["_gvis_cpu"] if batch["_gvis_cpu"] is not None else torch.as_tensor(batch["gmap_visited_masks"]).cpu().bool()
        g_vpids, vp_cand_vpids = batch["gmap_vpids"], batch["vp_cand_vpids"]

        # global branch (:146-150) and local branch (:159-162)
        gmap = Fn.EmbedAddF32.apply(self.P("gmap_step_embeddings.weight"), self._i32(g_step).view(-1), ops.h2d(g_img, dev, F32))
        gmap = Fn.add(gmap, self._seq2(ops.h2d(g_pos, dev, F32), "gmap_pos_embeddings"))
        vp = Fn.add(vp_img, self._seq2(ops.h2d(batch["vp_pos_fts"], dev, F32), "vp_pos_embeddings"))
        keep_g = (gm_cpu & ~gv_cpu)
        keep_g_dev = ops.h2d(keep_g.view(-1), dev, F32)
        gmap = Fn.RowScaleF32.apply(gmap.view(B * G, d), keep_g_dev)
        pm = ops.h2d(batch["pano_masks"], dev).to(F32).view(-1).contiguous()
        vp = Fn.RowScaleF32.apply(vp.view(B * Nv, d), pm)

        # host: which current-view candidate feeds which map slot (:174-190) -- one call into the C++ side-car on integer
        # node ids (a caller that already holds them passes `_gmap_ids` / `_cand_ids`; else the strings are interned here)
        if batch["_gmap_ids"] is not None:
            gi, ci = batch["_gmap_ids"], batch["_cand_ids"]
        else:
            gi, ci = graph.intern_vpids(g_vpids, vp_cand_vpids, G, Nv)
        src, inv, ttype = (torch.from_numpy(a) for a in graph.match_tables(gi, gv_cpu.numpy(), ci))
        fuse = Fn.GatherRowsF32.apply(vp, ops.h2d(src, dev), ops.h2d(inv, dev), gmap)
        fuse = Fn.EmbedAddF32.apply(self.P("token_type_embeddings.weight"), ops.h2d(ttype, dev), fuse)
        fuse = Fn.RowScaleF32.apply(fuse, keep_g_dev)                       # [B*G, d]

        cand_masks = keep_g
        cand_nums = cand_masks.sum(-1)
        # candidate permutation with the reference's CPU RNG call order (:216-223); the index tables around it
        # (selection in LM order, its inverse, the head column of every map slot) come from the side-car in one call
        perms = [torch.randperm(int(n) - 1) for n in cand_nums]
        sel, inv_sel, col = (torch.from_numpy(a) for a in graph.perm_tables(cand_masks.numpy(), [p.numpy() for p in perms]))
        cand_embeds = Fn.GatherRowsF32.apply(fuse, ops.h2d(sel, dev), ops.h2d(inv_sel, dev), None)

        hist_vis = self._stack_hist(batch["hist_vis"])
        ids, am, _ = self._tokens(batch, batch["prompts"])
        if self.kv is not None and not torch.is_grad_enabled() and self.kv.B == B:
            hk = self._hist_keys(batch["hist_vis"])
            Hs_cls = self._lm_cached(ids, am, cand_vis=cand_embeds, hist_vis=hist_vis, hist_keys=hk)
        elif torch.is_grad_enabled() and (self._episode_active() or (self.auto_episode and self.training and self.fp8 is None)):
            layout = self._vis_layout(ids, am, cand_embeds, hist_vis, None)
            if self._auto_open and not self._auto_matches(layout[0]):
                self._auto_close("next_episode")           # other prompts: the previous rollout's episode hands its gradients over
            if not self._episode_active():
                pre = self._auto_prefix_ids(batch, layout[0]) if (self.auto_episode and self.training) else None
                if pre is not None:
                    self.begin_episode(pre, teacher_forced=self.auto_form == "lazy")     # (flushes an accumulation window whose episodes are all finished)
                    self._auto_open = True
                    self.auto_stats["opened"] += 1
            Hs_cls = None
            if self._episode_active():
                Hs_cls = self._lm_episode(ids, am, cand_vis=cand_embeds, hist_vis=hist_vis, layout=layout)
            if Hs_cls is None:         # a left-truncated prompt (or no prefix to be told): this step takes the reference's formulation, gradients land in .grad at once
                Hs_cls = self._lm(ids, am, cand_vis=cand_embeds, hist_vis=hist_vis, cls_tail=True)
        else:
            Hs_cls = self._lm(ids, am, cand_vis=cand_embeds, hist_vis=hist_vis, cls_tail=True)
        if isinstance(Hs_cls, dict):
            # a step of a teacher-forced prefix-reuse episode: the LM forward, the head and the loss run in finish_episode()
            from .losses import DeferredLogits, LazyLogits
            Hs_cls["head"] = (ops.h2d(col, dev), ops.h2d(cand_masks.logical_not(), dev))
            # (an AUTOMATIC episode: the handle forces the step when its numbers are needed; an episode the caller opened with
            # teacher_forced=True promised not to read them: reading raises)
            handle = (LazyLogits if self._auto_open else DeferredLogits)(self.episode, Hs_cls, (B, G), dev, BF16)
            return {"fuse_embeds": fuse.detach().view(B, G, d), "fuse_logits": handle}
        pred = Fn.HeadBF16.apply(Hs_cls, self, "out_head.0")   # [B,100]

        # fuse_logits[b][cand slots] = [pred[b,0], pred[b,1:n][inv_perm]] ; -inf elsewhere (:234-242)
        logits = torch.gather(pred, 1, ops.h2d(col, dev)).masked_fill(ops.h2d(cand_masks.logical_not(), dev), float("-inf"))
        return {"fuse_embeds": fuse.detach().view(B, G, d), "fuse_logits": logits}

    # ------------------------------------------------------------------ object grounding (nav_model.py:407-451)
    def forward_object_grounding(self, mode, batch, training=True, **kwargs):
        dev, d = self.device, self.cfg.hidden_size
        obj_embeds, obj_masks = batch["obj_embeds"], torch.as_tensor(batch["obj_masks"]).cpu().bool()
        B, O, _ = obj_embeds.shape
        oe = Fn.add(obj_embeds, self._seq2(ops.h2d(batch["obj_loc_fts"], dev, F32), "obj_pos_embeddings")).view(B * O, d)
        cand_nums = obj_masks.sum(1) + 1
        sel = torch.nonzero(obj_masks.view(-1)).view(-1).to(torch.int32)
        inv = torch.full((B * O,), -1, dtype=torch.int32)
        inv[sel.long()] = torch.arange(sel.numel(), dtype=torch.int32)
        cand_vis = Fn.GatherRowsF32.apply(oe, ops.h2d(sel, dev), ops.h2d(inv, dev), None) if sel.numel() else None
        ids, am, _ = self._tokens(batch, batch["prompts"])
        if self.kv is not None and not torch.is_grad_enabled() and self.kv.B == B:
            # the grounding prompt of the last step shares instruction + history with the navigation prompts before it
            hk = self._hist_keys(batch["hist_vis"])
            Hs_cls = self._lm_cached(ids, am, cand_vis=cand_vis, hist_vis=self._stack_hist(batch["hist_vis"]), hist_keys=hk)
        else:
            Hs_cls = self._lm(ids, am, cand_vis=cand_vis, hist_vis=self._stack_hist(batch["hist_vis"]), cls_tail=True)
        pred = Fn.HeadBF16.apply(Hs_cls, self, "out_head.0")
        dead = torch.arange(pred.shape[1])[None] >= cand_nums[:, None]
        return {"obj_logits": pred.masked_fill(ops.h2d(dead, dev), float("-inf"))}

    # ------------------------------------------------------------------ LM-loss modes, training branches
    def _zero_pose_type0(self, rows):
        z = torch.zeros((rows, 14), dtype=F32, device=self.device)
        e = self._seq2(z, "vp_pos_embeddings")
        return Fn.add(e, self._Prow("token_type_embeddings.weight", 0))

    def forward_3dqa(self, mode, batch, training=True, **kwargs):
        """nav_model.py:346-404: training -> `.loss`; training=False -> greedy generation (`generated_sentences`)."""
        dev, d = self.device, self.cfg.hidden_size
        feats = [ops.h2d(f, dev, F32) for f in batch["features"]]
        B = len(feats)
        N = max(f.shape[0] for f in feats)
        vf = torch.zeros((B, N, feats[0].shape[1]), dtype=F32, device=dev)
        for i, f in enumerate(feats):
            vf[i, : f.shape[0]] = f
        vl = torch.tensor([f.shape[0] for f in feats])
        out = self.forward_panorama_per_step(vf, vl)
        pe = Fn.add(out["pano_embeds"].view(B * N, d), self._zero_pose_type0(B * N))
        pm = (torch.arange(N)[None] < vl[:, None])
        sel = torch.nonzero(pm.view(-1)).view(-1).to(torch.int32)
        inv = torch.full((B * N,), -1, dtype=torch.int32)
        inv[sel.long()] = torch.arange(sel.numel(), dtype=torch.int32)
        cand_vis = Fn.GatherRowsF32.apply(pe, ops.h2d(sel, dev), ops.h2d(inv, dev), None)
        if not training:
            ids, am, _ = self._tokens(batch, None if batch.get("input_ids") is not None else list(batch["prompts"]))
            return self._generate(ids, am, cand_vis, None, **kwargs)
        if batch.get("input_ids") is None:
            eos = self.lang_model.tokenizer.eos_token
            text = [[batch["prompts"][bn], batch["answers"][bn][0] + f"{eos}"] for bn in range(B)]
        else:
            text = None
        ids, am, tt = self._tokens(batch, text)
        labels = ids.clone()
        labels[tt[:, -labels.shape[-1]:] == 0] = -100
        Hs = self._lm(ids, am, cand_vis=cand_vis)
        return _Out(loss=self._lm_loss(Hs, ids, labels))

    def forward_summarization(self, mode, batch, training=True, **kwargs):
        """nav_model.py:251-343: training -> `.loss`; training=False -> greedy generation with `max_new_tokens=50` and
        the optional `trie` constraint (`generated_sentences`)."""
        dev, d = self.device, self.cfg.hidden_size
        vp_img = batch["vp_img_embeds"][:, 1:, :]
        nav_masks = torch.as_tensor(batch["vp_nav_masks"]).cpu().bool()[:, 1:]
        B, N, _ = vp_img.shape
        x = Fn.add(vp_img.contiguous().view(B * N, d), self._zero_pose_type0(B * N))
        sel = torch.nonzero(nav_masks.reshape(-1)).view(-1).to(torch.int32)
        inv = torch.full((B * N,), -1, dtype=torch.int32)
        inv[sel.long()] = torch.arange(sel.numel(), dtype=torch.int32)
        cand_vis = Fn.GatherRowsF32.apply(x, ops.h2d(sel, dev), ops.h2d(inv, dev), None) if sel.numel() else None
        if not training:
            ids, am, _ = self._tokens(batch, None if batch.get("input_ids") is not None else list(batch["prompts"]))
            return self._generate(ids, am, cand_vis, self._stack_hist(batch["hist_vis"]), max_new_tokens=50, trie=kwargs.get("trie"))
        if batch.get("input_ids") is None:
            eos = self.lang_model.tokenizer.eos_token
            dt = batch["data_type"][0]
            text = []
            for bn in range(B):
                label = (batch["answer"][bn] if dt in ("eqa", "fgr2r") else batch["instruction"][bn]) + f"{eos}"
                text.append([batch["prompts"][bn], label])
        else:
            text = None
        ids, am, tt = self._tokens(batch, text)
        labels = ids.clone()
        labels[tt[:, -labels.shape[-1]:] == 0] = -100
        Hs = self._lm(ids, am, cand_vis=cand_vis, hist_vis=self._stack_hist(batch["hist_vis"]))
        return _Out(loss=self._lm_loss(Hs, ids, labels))
