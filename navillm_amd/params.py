"""Parameter inventory of the NaviLLM hot path and seeded synthetic weights.

Key names and shapes are the reference's `state_dict()` layout (the
modules are declared at `models/nav_model.py:33-91`, `models/image_embedding.py:11-49`,
`models/detr_transformer.py:133-150`, HF `LlamaForCausalLM`) so that released
NaviLLM checkpoints load by name.  `tests/golden/make_golden.py` checks this
inventory against the reference module's own `state_dict()`.
"""
import zlib
import torch


def param_specs(cfg):
    """-> list of (name, shape, group) ; group in {'lm','f32'}.

    'lm' tensors take the LM dtype (bf16 under amp_bf16: modified_lm.py:47-48,
    nav_model.py:78-85), 'f32' tensors always stay fp32.
    """
    d, ff, V, L = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_layers
    h, ffe, F = cfg.enc_hidden_size, cfg.enc_intermediate_size, cfg.image_feat_size
    S = []
    S.append(("lang_model.model.embed_tokens.weight", (V, d), "lm"))
    for i in range(L):
        p = f"lang_model.model.layers.{i}."
        S += [
            (p + "self_attn.q_proj.weight", (d, d), "lm"),
            (p + "self_attn.k_proj.weight", (d, d), "lm"),
            (p + "self_attn.v_proj.weight", (d, d), "lm"),
            (p + "self_attn.o_proj.weight", (d, d), "lm"),
            (p + "mlp.gate_proj.weight", (ff, d), "lm"),
            (p + "mlp.up_proj.weight", (ff, d), "lm"),
            (p + "mlp.down_proj.weight", (d, ff), "lm"),
            (p + "input_layernorm.weight", (d,), "lm"),
            (p + "post_attention_layernorm.weight", (d,), "lm"),
        ]
    S.append(("lang_model.model.norm.weight", (d,), "lm"))
    S.append(("lang_model.lm_head.weight", (V, d), "lm"))

    e = "img_embeddings."
    S += [
        (e + "img_linear.weight", (h, F), "f32"), (e + "img_linear.bias", (h,), "f32"),
        (e + "img_layer_norm.weight", (h,), "f32"), (e + "img_layer_norm.bias", (h,), "f32"),
        (e + "loc_linear.weight", (h, cfg.angle_feat_size + 3), "f32"), (e + "loc_linear.bias", (h,), "f32"),
        (e + "loc_layer_norm.weight", (h,), "f32"), (e + "loc_layer_norm.bias", (h,), "f32"),
    ]
    if cfg.enable_og:
        if cfg.fuse_obj:
            S += [
                (e + "obj_linear.0.weight", (h, cfg.obj_feat_size), "f32"), (e + "obj_linear.0.bias", (h,), "f32"),
                (e + "obj_linear.1.weight", (h,), "f32"), (e + "obj_linear.1.bias", (h,), "f32"),
            ]
        S += [
            (e + "obj_projector.0.weight", (d, cfg.obj_feat_size), "f32"), (e + "obj_projector.0.bias", (d,), "f32"),
            (e + "obj_projector.1.weight", (d,), "f32"), (e + "obj_projector.1.bias", (d,), "f32"),
        ]
    S += [
        (e + "nav_type_embedding.weight", (3, h), "f32"),
        (e + "layer_norm.weight", (h,), "f32"), (e + "layer_norm.bias", (h,), "f32"),
    ]
    for i in range(cfg.num_pano_layers):
        p = e + f"pano_encoder.layers.{i}."
        S += [
            (p + "self_attn.in_proj_weight", (3 * h, h), "f32"), (p + "self_attn.in_proj_bias", (3 * h,), "f32"),
            (p + "self_attn.out_proj.weight", (h, h), "f32"), (p + "self_attn.out_proj.bias", (h,), "f32"),
            (p + "linear1.weight", (ffe, h), "f32"), (p + "linear1.bias", (ffe,), "f32"),
            (p + "linear2.weight", (h, ffe), "f32"), (p + "linear2.bias", (h,), "f32"),
            (p + "norm1.weight", (h,), "f32"), (p + "norm1.bias", (h,), "f32"),
            (p + "norm2.weight", (h,), "f32"), (p + "norm2.bias", (h,), "f32"),
        ]
    if cfg.num_pano_layers > 0:
        S += [(e + "pano_encoder.norm.weight", (h,), "f32"), (e + "pano_encoder.norm.bias", (h,), "f32")]
    S += [(e + "mapper.weight", (d, h), "f32"), (e + "mapper.bias", (d,), "f32")]

    S += [
        ("token_type_embeddings.weight", (3, d), "f32"),
        ("gmap_pos_embeddings.0.weight", (d, cfg.angle_feat_size + 3), "f32"), ("gmap_pos_embeddings.0.bias", (d,), "f32"),
        ("gmap_pos_embeddings.1.weight", (d,), "f32"), ("gmap_pos_embeddings.1.bias", (d,), "f32"),
        ("gmap_step_embeddings.weight", (cfg.max_action_steps, d), "f32"),
        ("vp_pos_embeddings.0.weight", (d, cfg.angle_feat_size * 2 + 6), "f32"), ("vp_pos_embeddings.0.bias", (d,), "f32"),
        ("vp_pos_embeddings.1.weight", (d,), "f32"), ("vp_pos_embeddings.1.bias", (d,), "f32"),
        ("obj_pos_embeddings.0.weight", (d, cfg.angle_feat_size + 3), "f32"), ("obj_pos_embeddings.0.bias", (d,), "f32"),
        ("obj_pos_embeddings.1.weight", (d,), "f32"), ("obj_pos_embeddings.1.bias", (d,), "f32"),
    ]
    if cfg.obj_feat_size > 0:
        S += [("og_head.0.weight", (100, d), "lm"), ("og_head.0.bias", (100,), "lm")]
    S += [("out_head.0.weight", (100, d), "lm"), ("out_head.0.bias", (100,), "lm")]
    return S


def _is_norm_weight(name):
    return (name.endswith("layernorm.weight") or name.endswith("norm.weight")
            or name.endswith("layer_norm.weight") or name.endswith("norm1.weight")
            or name.endswith("norm2.weight") or (name.endswith(".1.weight") and "embeddings" in name)
            or name.endswith("obj_projector.1.weight") or name.endswith("obj_linear.1.weight"))


def synth_tensor(name, shape, seed, device="cpu", std=0.02):
    """Deterministic per-name tensor (fp32). Norm scales ~ 1+0.1 N(0,1), biases and
    matrices ~ std*N(0,1); independent of generation order."""
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    t = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    if _is_norm_weight(name):
        return 1.0 + 0.1 * t
    if "embed" in name and len(shape) == 2 and "lang_model" not in name:
        return 0.05 * t
    return std * t


def synth_state_dict(cfg, seed=0, device="cpu"):
    """Seeded random weights with the reference dtypes (bf16 LM under amp_bf16)."""
    lm_dtype = torch.bfloat16 if cfg.lm_is_bf16 else torch.float32
    sd = {}
    for name, shape, group in param_specs(cfg):
        t = synth_tensor(name, shape, seed, device)
        sd[name] = t.to(lm_dtype) if group == "lm" else t
    return sd
