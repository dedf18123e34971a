"""Weights for the args-constructor path `NavModel(args, logger, model_config)` (models/nav_model.py:33-47):

* default (no `--resume_from_checkpoint`, no `--from_scratch`): the reference calls
  `ModifiedLlamaForCausalLM.from_pretrained(path)` and then `resize_token_embeddings` for the six added tokens
  (models/modified_lm.py:56-75).  `load_hf_llama` reads the same HF checkpoint directory (safetensors or .bin shards)
  straight into the flat HBM store: the first `base_vocab` rows of `embed_tokens` / `lm_head` come from the checkpoint, the
  added rows are drawn like HF's `_init_weights` does for a resized embedding (normal(0, initializer_range)).
* `--from_scratch` / `--resume_from_checkpoint`: the reference builds the LM from its config (`_init_weights`: Linear and
  Embedding ~ normal(0, initializer_range), RMSNorm = 1); a resumed run then overwrites everything through
  `tools/optims.py:12-24` (`NavModel.load_reference_state_dict`).  `reference_scratch_tensor` draws from the same
  distributions (not the same random stream -- nothing downstream depends on it).
* everything that is NOT the LM (scene encoder, fusion embeddings, heads) is freshly initialised by the reference in both
  cases with torch's module defaults; `reference_scratch_tensor` mirrors those defaults per parameter.

Nothing here silently falls back to synthetic weights: a missing or incomplete checkpoint raises."""
import glob
import json
import math
import os
import zlib

import torch


def iter_hf_tensors(path):
    """yield (name, tensor) for every tensor of a HF checkpoint directory (sharded or not; safetensors preferred)."""
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors import safe_open
        for fn in st:
            with safe_open(fn, framework="pt", device="cpu") as f:
                for k in f.keys():
                    yield k, f.get_tensor(k)
        return
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not bins:
        raise FileNotFoundError(
            f"{path}: no *.safetensors / pytorch_model*.bin weights. The args-constructor path loads the pretrained LM like the "
            "reference's from_pretrained(); pass from_scratch=True (or resume_from_checkpoint) for a from-config model, or "
            "build NavModel(nav_config=..., init='synthetic') for benchmarks.")
    for fn in bins:
        sd = torch.load(fn, map_location="cpu", weights_only=True)
        for k, v in sd.items():
            yield k, v


def _gen(name, seed, device="cpu"):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761) ^ 0x5bd1e995) & 0x7FFFFFFF)
    return g


def hf_llama_name(k):
    """HF LlamaForCausalLM key -> the reference's state_dict key (the LM sits at `NavModel.lang_model`)."""
    if k.endswith("rotary_emb.inv_freq"):
        return None
    return "lang_model." + k


@torch.no_grad()
def load_hf_llama(model, path, seed=0, initializer_range=None):
    """Fill the LM tensors of `model` (a navillm_amd NavModel) from the HF checkpoint directory `path`.
    -> number of tensors loaded.  Raises if a decoder tensor is missing or has the wrong shape."""
    cfg = model.cfg
    if initializer_range is None:
        try:
            with open(os.path.join(path, "config.json")) as f:
                initializer_range = json.load(f).get("initializer_range", 0.02)
        except OSError:
            initializer_range = 0.02
    want = {n for n in model._named if n.startswith("lang_model.")}
    seen = set()
    for k, v in iter_hf_tensors(path):
        n = hf_llama_name(k)
        if n is None or n not in want:
            continue
        dst = model._named[n]
        if n in ("lang_model.model.embed_tokens.weight", "lang_model.lm_head.weight"):
            rows = min(v.shape[0], cfg.base_vocab_size)
            if v.shape[1] != dst.shape[1] or v.shape[0] < cfg.base_vocab_size:
                raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} does not cover the {cfg.base_vocab_size} x {dst.shape[1]} base vocabulary")
            dst[:rows].copy_(v[:rows].to(dst.dtype))
            extra = dst.shape[0] - rows                      # <cand> <hist> <obj> <cls_1> <cls_2> <PAD>
            new = torch.randn((extra, dst.shape[1]), generator=_gen(n, seed), dtype=torch.float32) * initializer_range
            dst[rows:].copy_(new.to(dst.dtype))
        else:
            if tuple(v.shape) != tuple(dst.shape):
                raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(dst.shape)}")
            dst.copy_(v.to(dst.dtype))
        seen.add(n)
    missing = sorted(want - seen)
    if missing:
        raise KeyError(f"{path}: {len(missing)} LM tensors missing from the checkpoint, e.g. {missing[:4]}")
    return len(seen)


def reference_scratch_tensor(name, shape, seed=0, initializer_range=0.02, device="cpu"):
    """fp32 tensor drawn from the distribution the reference's constructor leaves this parameter with."""
    g = _gen(name, seed, device)
    kw = dict(generator=g, dtype=torch.float32, device=device)
    is_norm_w = (name.endswith("layernorm.weight") or name.endswith("norm.weight") or name.endswith("layer_norm.weight")
                 or name.endswith("norm1.weight") or name.endswith("norm2.weight")
                 or (name.endswith(".1.weight") and len(shape) == 1))
    if name.startswith("lang_model."):
        if len(shape) == 1:
            return torch.ones(shape, dtype=torch.float32, device=device)               # LlamaRMSNorm
        return torch.randn(shape, **kw) * initializer_range                              # _init_weights: Linear / Embedding
    if is_norm_w:
        return torch.ones(shape, dtype=torch.float32, device=device)                    # nn.LayerNorm
    if name.endswith(".bias") and (name.endswith("norm.bias") or name.endswith("layer_norm.bias") or name.endswith("norm1.bias")
                                   or name.endswith("norm2.bias") or (name.endswith(".1.bias") and "embeddings" in name)
                                   or name.endswith("obj_projector.1.bias") or name.endswith("obj_linear.1.bias")):
        return torch.zeros(shape, dtype=torch.float32, device=device)
    if name.endswith("in_proj_weight"):                                                  # nn.MultiheadAttention: xavier_uniform_
        a = math.sqrt(6.0 / (shape[0] + shape[1]))
        return (torch.rand(shape, **kw) * 2 - 1) * a
    if name.endswith("in_proj_bias") or name.endswith("self_attn.out_proj.bias"):
        return torch.zeros(shape, dtype=torch.float32, device=device)
    if len(shape) == 2 and "embedding" in name.split(".")[-2]:                          # nn.Embedding: N(0, 1)
        return torch.randn(shape, **kw)
    if len(shape) == 2:                                                                  # nn.Linear: kaiming_uniform_(a=sqrt(5))
        bound = 1.0 / math.sqrt(shape[1])
        return (torch.rand(shape, **kw) * 2 - 1) * bound
    return None                                                                          # Linear bias: needs fan_in, see below


@torch.no_grad()
def init_reference_scratch(model, seed=0, lm=True, rest=True):
    """draw every parameter of `model` (or only the LM / only the non-LM part) like the reference's constructor does"""
    from .params import param_specs
    shapes = {n: s for n, s, _ in param_specs(model.cfg)}
    dev = "cpu" if model.cfg.hidden_size <= 1024 else model.device
    for n, shape in shapes.items():
        is_lm = n.startswith("lang_model.")
        if (is_lm and not lm) or (not is_lm and not rest):
            continue
        t = reference_scratch_tensor(n, shape, seed, device=dev)
        if t is None:                                   # nn.Linear bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = shapes[n[:-4] + "weight"][1]
            t = (torch.rand(shape, generator=_gen(n, seed, dev), dtype=torch.float32, device=dev) * 2 - 1) / math.sqrt(fan_in)
        model.store.p(n).copy_(t)
