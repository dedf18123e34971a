"""GPU: weight-only fp8 (BASELINE config 5) -- navillm_amd/csrc/fp8w.hip + navillm_amd/fp8.py against
torch's float8_e4m3fn (the external definition of the format: fixture G11) and the fp8 oracle (oracle/navillm_oracle.py)."""
import types

import numpy as np
import pytest
import torch

from util import gold, T, tiny_cfg, meta_of, hist_lists, load_oracle, bf16_ulps_at_scale
from test_parity_gpu import build, maxerr, dev, pano_batch, _nav_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fp8_decode_and_quantiser_are_bit_exact_vs_torch_float8():
    from navillm_amd import fp8
    z = gold("g11_fp8_bf16.npz")
    tab = fp8.decode_table(torch.device(DEV)).float().cpu().numpy()
    want = z["decode_table"]
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(tab), np.isnan(want)), "NaN codes (0x7f, 0xff) differ: the hardware decode is not OCP e4m3fn"
    assert np.array_equal(tab[ok], want[ok]) and np.array_equal(np.signbit(tab[ok]), np.signbit(want[ok]))
    W = dev(z["quant_w"]).to(torch.bfloat16)
    q, s = fp8.quantize_rows(W)
    assert np.array_equal(q.cpu().numpy(), z["quant_codes"]), "e4m3fn codes differ from torch's conversion"
    assert np.array_equal(s.cpu().numpy(), z["quant_scales"])
    assert torch.equal(fp8.dequantize_rows(q, s).float().cpu(), T(z["quant_dequant"]).float())
    # a larger random matrix, incl. values that land exactly on rounding ties
    torch.manual_seed(0)
    Wl = (torch.randn(300, 1024) * torch.rand(300, 1) * 3).to(torch.bfloat16)
    Wl[:, ::7] = (Wl[:, ::7].float() * 0.5).to(torch.bfloat16)
    O = load_oracle()
    qo, so = O.fp8_quantize_rows(Wl)
    q, s = fp8.quantize_rows(Wl.to(DEV))
    assert torch.equal(q.cpu(), qo.view(torch.uint8)) and torch.equal(s.cpu(), so)


@pytest.mark.parametrize("M,N,K,resid", [(8, 512, 256, False), (3, 1000, 1408, True), (16, 13824, 5120, True), (1, 64, 64, False),
                                         (8, 27648, 5120, False), (5, 5120, 13824, True), (12, 22016, 4096, False), (8, 4096, 4096, True),
                                         (2, 8, 128, False)])
def test_gemv_fp8w_matches_dequantised_gemm(M, N, K, resid):
    """decode-step weight streamer on the codes == bf16 GEMM on the de-quantised operand == fp32 reference within one rounding"""
    from navillm_amd import fp8, ops
    torch.manual_seed(1)
    W = (torch.randn(N, K) * 0.05).to(torch.bfloat16).to(DEV)
    x = torch.randn(M, K).to(torch.bfloat16).to(DEV)
    R = torch.randn(M, N).to(torch.bfloat16).to(DEV) if resid else None
    q, s = fp8.quantize_rows(W)
    Wd = fp8.dequantize_rows(q, s)
    epi = ops.EPI_RESID if resid else ops.EPI_STORE
    got = fp8.gemv_fp8w(x, q, s, R=R, epilogue=epi).float().cpu()
    ops.GEMV_DECODE = False
    try:
        tile = ops.gemm_bf16(ops.NT, x, Wd, R=R, epilogue=epi).float().cpu()
    finally:
        ops.GEMV_DECODE = True
    ref = x.float().cpu() @ Wd.float().cpu().T
    want = (R.float().cpu() + ref.to(torch.bfloat16).float()).to(torch.bfloat16).float() if resid else ref.to(torch.bfloat16).float()
    ulp = 2.0 ** (np.floor(np.log2(want.abs().max().item())) - 7)
    assert (got - want).abs().max().item() <= ulp and (got - tile).abs().max().item() <= ulp
    assert ((got - want).abs() > 0).float().mean().item() < 0.05       # only accumulation-order flips of the final rounding


@pytest.mark.parametrize("gemm_mode", [7, 9])
def test_fp8_weight_only_navigation_and_generation_vs_reference_on_dequantised_weights(gemm_mode):
    """end to end on the G3 / G9 inputs: the model after to_fp8_weight_only() against the REFERENCE run on de-quantised weights
    (fixture G11), through the full-recompute path and the K/V cache; greedy generation (decode steps = nv_gemv_fp8w) against
    the oracle on de-quantised weights; memory; inference-only."""
    from test_oracle_golden import g9_inputs
    from navillm_amd.params import synth_state_dict
    O = load_oracle()
    z11, z3 = gold("g11_fp8_bf16.npz"), gold("g3_nav_bf16.npz")
    cfg = tiny_cfg("bf16")
    m = build(cfg)
    lm_bf16_bytes = m.store.param["lm"].numel() * 2
    f8 = m.to_fp8_weight_only(gemm_mode=gemm_mode)
    # mode 7 (default) IS the fixture's semantics (operands bf16(s*q)); mode 9 applies s to the fp32 accumulator -- one bf16 rounding per
    # weight fewer -- and is allowed one more output spacing (measured 2.0 / 3.0 where mode 7 measures 2.0 / 2.5)
    bound = 2.5 if gemm_mode == 7 else 3.5
    dec = sum(2 * s[0] * s[1] for n, s in m.store.shape_of.items() if n in m.store.released)
    assert f8.bytes <= 0.51 * dec + 4 * 4096 and m.store.param["lm"].numel() * 2 == lm_bf16_bytes - dec     # codes + scales; bf16 copies gone
    assert m.store.grad is None and m.P("lang_model.model.layers.0.self_attn.q_proj.weight").numel() == 0
    l16 = T(z11["fuse_logits"])
    with torch.no_grad():
        _, out, _ = _nav_forward(m, z3)
    u = bf16_ulps_at_scale(out["fuse_logits"], l16)
    gap_unq = maxerr(out["fuse_logits"], T(z3["fuse_logits"]))
    print(f"[fp8 g11] logits vs reference-on-dequantised-weights: {maxerr(out['fuse_logits'], l16):.5f} = {u:.2f} bf16 ulps "
          f"(distance to the UNquantised reference: {gap_unq:.4f})")
    assert u <= bound and gap_unq > 4 * maxerr(out["fuse_logits"], l16)
    # through the K/V cache (prefill GEMMs on the de-quantised scratch panel)
    m.enable_kv_cache(3)
    with torch.no_grad():
        _, outc, _ = _nav_forward(m, z3)
    uc = bf16_ulps_at_scale(outc["fuse_logits"], l16)
    print(f"[fp8 g11 mode {gemm_mode}] through the K/V cache: {uc:.2f} bf16 ulps")
    assert uc <= bound
    m.kv = None
    # generation: decode steps run on the codes (nv_gemv_fp8w); oracle = greedy recompute on de-quantised weights
    z = gold("g9_generate_bf16.npz")
    meta, feats_cpu, trie = g9_inputs(z)
    Pq = O.fp8_weight_only_state_dict(synth_state_dict(cfg, 11))
    with torch.no_grad():
        want = O.qa_3d_generate(Pq, cfg, feats_cpu, T(z["qa_input_ids"]), T(z["qa_attention_mask"]), max_new_tokens=6,
                                eos_token_id=meta["eos"], pad_token_id=meta["pad"])
        m.lang_model.tokenizer = types.SimpleNamespace(eos_token_id=meta["eos"], unk_token_id=meta["pad"])
        got = m("3dqa", dict(features=[f.to(DEV) for f in feats_cpu], question=["q"] * 3, input_ids=T(z["qa_input_ids"]),
                             attention_mask=T(z["qa_attention_mask"])), training=False, do_sample=False, max_new_tokens=6)
    assert got["generated_ids"] == want, (got["generated_ids"], want)
    # inference only
    with pytest.raises(RuntimeError, match="inference only"):
        _nav_forward(m, z3)


@pytest.mark.parametrize("gemm_mode", [9, 7])
def test_fp8_13b_shaped_layer_vs_fp8_oracle(gemm_mode):
    """(gemm_mode: how nv_gemm_fp8w consumes the codes -- 9 = scale on the fp32 accumulator (default), 7 = operands bf16(s*q).)
    BASELINE config 5's layer shape (Vicuna-13B: d=5120, 40 heads, ff=13824) with weight-only fp8, one decoder layer, B=4:
    prefill GEMMs on the de-quantised scratch panel (20 / 54 / 108 column tiles) and the pruned-tail rows through
    nv_gemv_fp8w -- against the oracle on de-quantised weights, in bf16 and fp32."""
    from navillm_amd import config as nvcfg
    from navillm_amd.nav_model import NavModel
    from navillm_amd.params import synth_state_dict
    from navillm_amd.synthetic import SyntheticEpisodes
    O = load_oracle()
    cfg = nvcfg.NavConfig(hidden_size=5120, num_layers=1, num_heads=40, intermediate_size=13824, base_vocab_size=1000,
                          enc_hidden_size=256, enc_num_heads=4, enc_intermediate_size=512, image_feat_size=768)
    m = NavModel(nav_config=cfg, device=torch.device(DEV), seed=5)
    m.eval()
    P16 = synth_state_dict(cfg, 5)
    with torch.no_grad():
        assert m.load_reference_state_dict(P16) == len(P16)
    m.to_fp8_weight_only(gemm_mode=gemm_mode)
    Pq16 = O.fp8_weight_only_state_dict(P16)
    Pq32 = {k: v.float() for k, v in Pq16.items()}
    cfg32 = nvcfg.NavConfig(**{**cfg.__dict__, "precision": "fp32"})
    ep = SyntheticEpisodes(cfg, 4, seed=77, instr_len=150, device=torch.device(DEV))
    pin = ep.panorama_inputs()
    with torch.no_grad():
        pano = m("panorama", pin)
    ep.update_maps(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    nav = ep.nav_inputs(pano["pano_embeds"], pano["pano_masks"], pin["cand_vpids"])
    ids, am = ep.tokenise(nav, "<cls_1>")
    nav["input_ids"], nav["attention_mask"] = ids, am
    torch.manual_seed(100)
    with torch.no_grad():
        out = m("navigation", nav)
    cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in nav.items()}
    refs = {}
    for tag, P, c in (("bf16", Pq16, cfg), ("fp32", Pq32, cfg32)):
        torch.manual_seed(100)
        with torch.no_grad():
            refs[tag] = O.navigation(P, c, cpu, ids, am)["fuse_logits"]
    torch.manual_seed(100)
    with torch.no_grad():
        unq = O.navigation({k: v.float() for k, v in P16.items()}, cfg32, cpu, ids, am)["fuse_logits"]
    e_hip, e_ref = maxerr(out["fuse_logits"], refs["fp32"]), maxerr(refs["bf16"], refs["fp32"])
    dq = maxerr(refs["fp32"], unq)
    scale = float(refs["bf16"][torch.isfinite(refs["bf16"])].abs().max())
    ulp = 2.0 ** (int(np.floor(np.log2(scale))) - 7)
    print(f"[fp8 13b-layer] S={ids.shape[1]} |hip-orc32(dequant)|={e_hip:.5f} vs |orc16-orc32|={e_ref:.5f} (ratio {e_hip / e_ref:.2f}); "
          f"quantisation itself moves the fp32 logits by {dq:.4f} (scale {scale:.2f})")
    assert e_hip <= 1.25 * e_ref + ulp


def test_fp8_resident_bf16_operands_equal_the_per_call_dequantisation():
    """to_fp8_weight_only(resident_bf16=True): prefill / K/V-reuse GEMMs read the kept bf16(s*q) operands, decode steps the codes --
    bit-identical to the memory-lean form (same de-quantised values), before and after fp8_release_resident()"""
    z3 = gold("g3_nav_bf16.npz")
    cfg = tiny_cfg("bf16")
    lean, res = build(cfg), build(cfg)
    lean.to_fp8_weight_only()
    f8 = res.to_fp8_weight_only(resident_bf16=True)
    assert f8.resident is not None and res.store.grad is None and res.P("lang_model.model.layers.0.self_attn.q_proj.weight").numel() > 0
    outs = {}
    for tag, m in (("lean", lean), ("resident", res)):
        with torch.no_grad():
            _, o, _ = _nav_forward(m, z3)
            m.enable_kv_cache(3)
            _, oc, _ = _nav_forward(m, z3)
            m.kv = None
        outs[tag] = (o["fuse_logits"].clone(), oc["fuse_logits"].clone())
    assert torch.equal(outs["lean"][0], outs["resident"][0]) and torch.equal(outs["lean"][1], outs["resident"][1])
    res.fp8_release_resident()
    assert f8.resident is None and res.P("lang_model.model.layers.0.self_attn.q_proj.weight").numel() == 0
    with torch.no_grad():
        _, o2, _ = _nav_forward(res, z3)
    assert torch.equal(o2["fuse_logits"], outs["lean"][0])


def test_fp8_overlapped_dequantisation_equals_the_in_line_one(monkeypatch):
    """round 3: the next Linear's operand is de-quantised on a side stream while the current GEMM runs (two panels; Python path:
    navillm_amd/fp8.py, K/V-cache path: nv_decoder_set_fp8_overlap).  Same values, same GEMMs -> bit-identical to the in-line
    pre-pass, across repeated forwards (the prefetch wraps to layer 0) and with the two paths interleaved (they share the panels)."""
    z3 = gold("g3_nav_bf16.npz")
    cfg = tiny_cfg("bf16")
    monkeypatch.setenv("NAVILLM_FP8_OVERLAP", "0")
    inline = build(cfg)
    f0 = inline.to_fp8_weight_only()
    monkeypatch.setenv("NAVILLM_FP8_OVERLAP", "1")
    over = build(cfg)
    f1 = over.to_fp8_weight_only()
    assert not f0.overlap and f1.overlap
    seq = ["full", "full", "kv", "full", "kv", "kv", "full"]
    outs = {}
    for tag, m in (("inline", inline), ("overlap", over)):
        got = []
        with torch.no_grad():
            for step in seq:
                if step == "kv":
                    m.enable_kv_cache(3)
                _, o, _ = _nav_forward(m, z3)
                m.kv = None
                got.append(o["fuse_logits"].clone())
        torch.cuda.synchronize()
        outs[tag] = got
    assert f1._panels is not None and f0._panels is None
    for a, b in zip(outs["inline"], outs["overlap"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K,tile,resid", [(800, 5120, 5120, 85, False), (500, 15360, 5120, 84, True), (640, 4096, 11008, 85, False),
                                              (670, 1024, 512, 84, True), (130, 776, 192, 84, False), (800, 27648, 5120, 0, False),
                                              (900, 5120, 13824, 0, True)])
def test_tile_gemm_on_fp8_codes_equals_prepass_then_bf16_gemm(M, N, K, tile, resid):
    """round 4 ("fp8 codes into the GEMM"): nv_gemm_fp8w DMAs the weight tile as e4m3fn bytes and converts it to
    bf16(s * q) on the MFMA fragment path (mode 7) -- the SAME operand values, K order and tile shape as nv_fp8_dequant_rows followed
    by the bf16 GEMM on that tile, so the results must be BIT-IDENTICAL (ragged N, K-slices of the split-K tail, residual epilogue
    included).  Modes 8 / 9 (one v_cvt_scalef32_pk_bf16_fp8 per pair; the scale as the instruction's operand / on the fp32
    accumulator) are measured here too: 9 must stay within two output spacings, 8 is reported (whether the instruction honours the
    scale's mantissa is what this run finds out)."""
    from navillm_amd import fp8, ops
    torch.manual_seed(3)
    W = (torch.randn(N, K) * 0.05 * (0.25 + torch.rand(N, 1) * 4)).to(torch.bfloat16).to(DEV)     # row scales over a 16x range
    x = torch.randn(M, K).to(torch.bfloat16).to(DEV)
    R = torch.randn(M, N).to(torch.bfloat16).to(DEV) if resid else None
    q, s = fp8.quantize_rows(W)
    Wd = fp8.dequantize_rows(q, s)
    epi = ops.EPI_RESID if resid else ops.EPI_STORE
    got7 = fp8.gemm_fp8w(x, q, s, R=R, epilogue=epi, mode=7, tile_cfg=tile)
    if tile == 0 and got7 is None:
        pytest.skip("the planner does not pick a cut-off tile for this shape on this build")
    assert got7 is not None
    cfg = tile
    if tile == 0:                                        # whichever of the two the planner chose: bit-identical to one of them
        refs = [ops.gemm_bf16(ops.NT, x, Wd, R=R, epilogue=epi, tile_cfg=c) for c in (84, 85)]
        assert any(torch.equal(got7, r) for r in refs), "mode 7 differs from the pre-pass + bf16 GEMM on both cut-off tiles"
        ref = refs[0] if torch.equal(got7, refs[0]) else refs[1]
    else:
        ref = ops.gemm_bf16(ops.NT, x, Wd, R=R, epilogue=epi, tile_cfg=cfg)
        assert torch.equal(got7, ref), f"mode 7: {(got7.float() - ref.float()).abs().max().item()} max abs diff"
    scale = ref.float().abs().max().item()
    ulp = 2.0 ** (np.floor(np.log2(scale)) - 7)
    got9 = fp8.gemm_fp8w(x, q, s, R=R, epilogue=epi, mode=9, tile_cfg=tile)
    got8 = fp8.gemm_fp8w(x, q, s, R=R, epilogue=epi, mode=8, tile_cfg=tile)
    d9 = (got9.float() - ref.float()).abs().max().item()
    d8 = (got8.float() - ref.float()).abs().max().item()
    print(f"[gemm_fp8w M={M} N={N} K={K} tile={tile}] mode 7 bit-identical; mode 9 (scale on the accumulator) max diff {d9 / ulp:.2f} output spacings, "
          f"{(got9 != ref).float().mean().item():.3%} of the elements differ; mode 8 (scale operand of v_cvt_scalef32) max diff {d8 / ulp:.2f} spacings, "
          f"bit-identical: {torch.equal(got8, ref)}")
    assert d9 <= 2 * ulp


def test_tile_gemm_on_fp8_codes_plans_per_shape_and_validates_arguments():
    """the entry point decides per shape (launch model: its 128 / 160-row tile against the pre-pass + the bf16 GEMM's best tile) and says
    "not my shape" (None) where the alternative is estimated faster; whatever it accepts in mode 7 equals the pre-pass path on that
    tile bit for bit; invalid arguments raise"""
    from navillm_amd import fp8, ops
    torch.manual_seed(3)
    W = (torch.randn(4096, 4096) * 0.05).to(torch.bfloat16).to(DEV)
    q, s = fp8.quantize_rows(W)
    Wd = fp8.dequantize_rows(q, s)
    served = {}
    for M in (640, 4096, 5152):
        x = torch.randn(M, 4096).to(torch.bfloat16).to(DEV)
        y = fp8.gemm_fp8w(x, q, s, mode=7)
        served[M] = y is not None
        if y is not None:
            assert any(torch.equal(y, ops.gemm_bf16(ops.NT, x, Wd, tile_cfg=c)) for c in (84, 85))
    print(f"[gemm_fp8w N=K=4096, mode 7] served by the fp8 tile kernel: {served}")
    assert served[640]
    x = torch.randn(640, 4096).to(torch.bfloat16).to(DEV)
    with pytest.raises(Exception):
        fp8.gemm_fp8w(x, q, s, epilogue=1)
    with pytest.raises(Exception):
        fp8.gemm_fp8w(x, q, s, mode=5)
    prev = ops._L().nv_gemm_fp8w_default_mode(0)
    assert prev in (7, 9) and ops._L().nv_gemm_fp8w_default_mode(7) == prev and ops._L().nv_gemm_fp8w_default_mode(prev) == 7
