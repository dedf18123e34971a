"""Shape/config record for the NaviLLM hot path.

Mirrors what the reference derives in `models/nav_model.py:17-29` (init_vis_config:
encoder hyper-parameters come from bert-large-uncased) and what it reads from the
LM's HF config (`models/nav_model.py:40-47`, `models/modified_lm.py:38-75`).
Nothing here depends on `transformers`.
"""
from dataclasses import dataclass, asdict
import json
import os


@dataclass
class NavConfig:
    # ---- causal LM (Llama family; reference: HF LlamaConfig) ----
    hidden_size: int = 4096          # d
    num_layers: int = 32             # L
    num_heads: int = 32              # H  (head_dim = d / H, the HIP attention is built for 128)
    intermediate_size: int = 11008   # ff
    base_vocab_size: int = 32000     # tokenizer vocab before the 6 added tokens
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    # ---- scene encoder (reference: bert-large-uncased fields, nav_model.py:18-19) ----
    enc_hidden_size: int = 1024      # h
    enc_num_heads: int = 16
    enc_intermediate_size: int = 4096
    num_pano_layers: int = 2         # configs/multi.yaml Model.num_pano_layers
    image_feat_size: int = 1024      # F (768 for the BASELINE synthetic config)
    angle_feat_size: int = 4
    obj_feat_size: int = 768
    max_action_steps: int = 100
    # ---- switches (args.*) ----
    enable_og: bool = True           # use_obj  -> obj_projector exists
    fuse_obj: bool = False
    feat_dropout: float = 0.4
    enc_dropout: float = 0.1
    precision: str = "amp_bf16"      # 'fp32' | 'amp_bf16'  (modified_lm.py:40-46)

    # added tokens, in the order init_tokenizer adds them (modified_lm.py:59-66)
    @property
    def cand_token_id(self):
        return self.base_vocab_size + 0

    @property
    def hist_token_id(self):
        return self.base_vocab_size + 1

    @property
    def obj_token_id(self):
        return self.base_vocab_size + 2

    @property
    def cls_token_ids(self):
        return (self.base_vocab_size + 3, self.base_vocab_size + 4)

    @property
    def pad_token_id(self):
        return self.base_vocab_size + 5

    @property
    def special_token_ids(self):
        return tuple(range(self.base_vocab_size, self.base_vocab_size + 5))

    @property
    def vocab_size(self):
        return self.base_vocab_size + 6

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def lm_is_bf16(self):
        return "bf16" in self.precision or "bfloat16" in self.precision

    def to_json(self):
        return json.dumps(asdict(self), indent=1, sort_keys=True)

    @staticmethod
    def from_json(s):
        return NavConfig(**json.loads(s))

    @staticmethod
    def from_hf_dir(path, **over):
        """Read a HF `config.json` (LlamaConfig) without importing transformers."""
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        kw = dict(
            hidden_size=c["hidden_size"], num_layers=c["num_hidden_layers"],
            num_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
            base_vocab_size=c["vocab_size"], rms_norm_eps=c.get("rms_norm_eps", 1e-6),
        )
        rp = c.get("rope_parameters") or {}
        kw["rope_theta"] = c.get("rope_theta", rp.get("rope_theta", 10000.0))
        kw.update(over)
        return NavConfig(**kw)


def vicuna_7b(**over):
    kw = dict(hidden_size=4096, num_layers=32, num_heads=32, intermediate_size=11008,
              base_vocab_size=32000)
    kw.update(over)
    return NavConfig(**kw)


def vicuna_13b(**over):
    kw = dict(hidden_size=5120, num_layers=40, num_heads=40, intermediate_size=13824,
              base_vocab_size=32000)
    kw.update(over)
    return NavConfig(**kw)


def tiny(**over):
    """Plumbing-size model used by the golden fixtures (tests/golden/make_golden.py)."""
    kw = dict(hidden_size=256, num_layers=2, num_heads=2, intermediate_size=512,
              base_vocab_size=250, enc_hidden_size=128, enc_num_heads=4,
              enc_intermediate_size=256, image_feat_size=64, obj_feat_size=48)
    kw.update(over)
    return NavConfig(**kw)
