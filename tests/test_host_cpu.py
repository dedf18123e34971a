"""CPU: host-side logic and the C-ABI surface (no kernel is launched here)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from util import gold, meta_of, ROOT, GOLD, tiny_cfg


def test_library_exports_every_declared_symbol():
    """include/navillm_hip.h <-> navillm_amd/lib.py <-> the built .so agree symbol for symbol."""
    from navillm_amd import lib
    hdr = open(os.path.join(ROOT, "include", "navillm_hip.h")).read()
    declared = set(re.findall(r"\b(nv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    if not os.path.exists(lib.LIB_PATH):
        from navillm_amd import build
        build.build(verbose=False)
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), name
    # cheap calls that do not touch a device
    assert L.nv_rmsnorm_bwd_workspace_bytes(4096) == 1024 * 4096 * 4
    assert L.nv_attn_bwd_workspace_bytes(2, 10, 3) == 2 * 10 * 3 * 4
    assert L.nv_layernorm_bwd_workspace_bytes(1024) == 2 * 128 * 1024 * 4
    # argument validation happens before any launch
    assert L.nv_gemm_bf16(0, None, None, None, None, 4, 4, 64, 64, 64, 4, 0, 0, 0, None) == -1


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    from navillm_amd import lib
    from navillm_amd.nav_model import NavModel
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            NavModel(nav_config=tiny_cfg("bf16"), device=torch.device("cpu"))
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libnavillm_hip.so")
    with pytest.raises(lib.NaviLLMHipError, match="no CPU/PyTorch fallback"):
        lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "navillm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "navillm_oracle" not in src and "import oracle" not in src and "from oracle" not in src, fn


def test_flat_store_layout():
    from navillm_amd.flat import FlatStore, ALIGN
    from navillm_amd.params import param_specs
    cfg = tiny_cfg("bf16")
    st = FlatStore(cfg, "cpu")
    d, ff = cfg.hidden_size, cfg.intermediate_size
    for n, shape, grp in param_specs(cfg):
        assert st.offsets[n] % ALIGN == 0
        assert tuple(st.p(n).shape) == tuple(shape) and st.p(n).dtype == (torch.bfloat16 if grp == "lm" else torch.float32)
        assert st.p(n).data_ptr() % 16 == 0
    for i in range(cfg.num_layers):
        q = st.p(f"lang_model.model.layers.{i}.self_attn.q_proj.weight")
        v = st.p(f"lang_model.model.layers.{i}.self_attn.v_proj.weight")
        P = st.qkv(i)
        assert P.shape == (3 * d, d) and P.data_ptr() == q.data_ptr() and P[2 * d:].data_ptr() == v.data_ptr()
        g = st.gate_up(i)
        assert g.shape == (2 * ff, d) and g[ff:].data_ptr() == st.p(f"lang_model.model.layers.{i}.mlp.up_proj.weight").data_ptr()
        s, e = st.layer_slice(i)
        assert st.offsets[f"lang_model.model.layers.{i}.self_attn.q_proj.weight"] == s
        if i + 1 < cfg.num_layers:
            assert st.layer_slice(i + 1)[0] == e
    lp = st.lm_head_padded()
    assert lp.shape[0] % 64 == 0 and lp.shape[0] >= cfg.vocab_size
    # writes through a parameter view land in the flat buffer (what nn.Parameter views rely on)
    st.p("out_head.0.bias").fill_(3.0)
    o = st.offsets["out_head.0.bias"]
    assert float(st.param["lm"][o]) == 3.0


def test_g6_prompt_strings():
    from navillm_amd.prompts import navigation_prompt
    g = json.load(open(os.path.join(GOLD, "g6_prompts.json")))
    n = 0
    for k, v in g.items():
        agent, mode, h, c = k.split("/")
        if mode == "navigation" and agent in ("r2r", "reverie"):
            assert navigation_prompt(agent, "INSTR", int(h), int(c)) == v
            n += 1
    assert n >= 6


def test_g7_graph_sidecar_matches_reference():
    from navillm_amd.graph import FloydGraph, GraphMap, calculate_vp_rel_pos_fts, get_angle_fts
    z = gold("g7_graph.npz")
    pos = {f"n{i}": z["positions"][i] for i in range(7)}
    fg = FloydGraph()
    for a, b in z["edges"]:
        fg.add_edge(f"n{a}", f"n{b}", float(np.linalg.norm(pos[f"n{a}"] - pos[f"n{b}"])))
    for s, k in enumerate(("n1", "n2", "n4", "n5")):
        fg.update(k)
        got = np.array([[fg.distance(f"n{i}", f"n{j}") for j in range(7)] for i in range(7)], dtype=np.float64)
        assert np.allclose(got, z["dists_after"][s], rtol=0, atol=1e-9), s
    paths = meta_of(z)["paths"]
    for key, want in paths.items():
        i, j = key.split("-")
        assert fg.path(f"n{i}", f"n{j}") == want, key
    h, e, dd = calculate_vp_rel_pos_fts(pos["n0"], np.stack([pos[f"n{j}"] for j in range(1, 7)]), 0.3, -0.1)
    assert np.allclose(np.stack([h, e, dd], 1), z["rel"], atol=1e-12)
    assert np.array_equal(get_angle_fts(z["rel"][:, 0], z["rel"][:, 1], 4), z["ang"])
    gm = GraphMap("n0")
    gm.node_positions = dict(pos)
    gm.graph = fg
    pf = gm.get_pos_fts("n1", [None, "n0", "n2", "n3", "n5"], 0.3, -0.1)
    assert np.allclose(pf, z["pos_fts"], atol=1e-6)


def test_stub_tokenizer_and_episode_shapes():
    """host side of the synthetic driver: layout of ids (<hist> x t, <cand> x K, <cls_1> last, left pad)."""
    from navillm_amd.synthetic import StubTokenizer
    from navillm_amd.prompts import navigation_prompt
    cfg = tiny_cfg("bf16")
    tok = StubTokenizer(cfg)
    seqs = []
    for t, k, n in ((0, 3, 5), (2, 6, 9)):
        p = navigation_prompt("r2r", "{INSTR}", t, k, "<cls_1>")
        ids = tok.encode(p, list(range(10, 10 + n)))
        assert ids[0] == 1 and ids[-1] == cfg.cls_token_ids[0]
        assert ids.count(cfg.hist_token_id) == t and ids.count(cfg.cand_token_id) == k - 1
        assert all(0 <= i < cfg.vocab_size for i in ids)
        seqs.append(ids)
    ids, am = tok.pad_left(seqs)
    assert ids.shape == am.shape and bool((ids[:, -1] == cfg.cls_token_ids[0]).all())
    pad = ids.shape[1] - len(seqs[0])
    assert bool((ids[0, :pad] == cfg.pad_token_id).all()) and int(am[0].sum()) == len(seqs[0])
    long = tok.pad_left([list(range(3, 3 + 2000))])[0]
    assert long.shape[1] == 1024 and int(long[0, -1]) == 2002      # left truncation keeps the tail


def test_config_from_hf_dir_and_special_ids():
    from navillm_amd.config import NavConfig
    cfg = NavConfig.from_hf_dir(os.path.join(GOLD, "tiny_llama"), enc_hidden_size=128)
    t = tiny_cfg("bf16")
    assert (cfg.hidden_size, cfg.num_layers, cfg.num_heads, cfg.intermediate_size, cfg.base_vocab_size) == \
        (t.hidden_size, t.num_layers, t.num_heads, t.intermediate_size, t.base_vocab_size)
    assert cfg.special_token_ids == tuple(range(250, 255)) and cfg.pad_token_id == 255 and cfg.vocab_size == 256
