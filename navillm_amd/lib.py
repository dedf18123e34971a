"""ctypes binding of libnavillm_hip.so -- the ONLY way the package reaches the GPU kernels.

There is no fallback: if the library is missing or a symbol declared in include/navillm_hip.h is
not exported, importing/using the product path raises.  (The CPU oracle lives in oracle/ and is
never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnavillm_hip.so")

vp = ip = fp = lp = C.c_void_p   # every device pointer travels as void*
i, f, l, sz = C.c_int, C.c_float, C.c_long, C.c_size_t

# name -> (restype, argtypes); mirrors include/navillm_hip.h one to one
SIGNATURES = {
    "nv_gemm_bf16": (i, [i, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp]),
    "nv_gemm_bf16_workspace_bytes": (sz, []),
    "nv_gemm_bf16_ws": (i, [i, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, vp]),
    "nv_gemm_bf16_rope": (i, [vp, vp, vp, vp, vp, ip, i, i, i, i, i, i, i, i, vp, vp]),
    "nv_gemm_bf16_rope_cfg": (i, [vp, vp, vp, vp, vp, ip, i, i, i, i, i, i, i, i, i, vp, vp]),
    "nv_gemv_bf16": (i, [vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp]),
    "nv_fp8_quant_rows": (i, [vp, vp, fp, i, i, i, i, vp]),
    "nv_fp8_dequant_rows": (i, [vp, fp, vp, i, i, i, i, vp]),
    "nv_fp8_decode_table": (i, [vp, vp]),
    "nv_gemv_fp8w": (i, [vp, vp, fp, vp, vp, i, i, i, i, i, i, i, i, vp]),
    "nv_gemm_fp8w": (i, [vp, vp, fp, vp, vp, i, i, i, i, i, i, i, i, i, i, vp, vp]),
    "nv_gemm_fp8w_default_mode": (i, [i]),
    "nv_embed_vis_bf16": (i, [vp, ip, ip, fp, vp, i, i, vp]),
    "nv_vis_grad_f32": (i, [vp, ip, fp, i, i, vp]),
    "nv_embed_grad_bf16": (i, [vp, ip, ip, ip, vp, i, i, vp]),
    "nv_rmsnorm_fwd_bf16": (i, [vp, vp, vp, fp, i, i, f, vp]),
    "nv_rmsnorm_bwd_workspace_bytes": (sz, [i]),
    "nv_rmsnorm_bwd_bf16": (i, [vp, vp, vp, fp, vp, vp, vp, vp, i, i, vp]),
    "nv_rope_bf16": (i, [vp, vp, vp, i, i, i, i, i, i, vp]),
    "nv_rope_rows_bf16": (i, [vp, vp, vp, ip, i, i, i, i, vp]),
    "nv_rope_rows_t_bf16": (i, [vp, vp, vp, ip, i, i, i, i, vp]),
    "nv_kv_grad_accum_f32": (i, [vp, fp, ip, i, i, vp]),
    "nv_kv_grad_set_f32": (i, [vp, fp, ip, i, i, vp]),
    "nv_kv_grad_inject_bf16": (i, [vp, fp, ip, i, i, vp]),
    "nv_swiglu_fwd_bf16": (i, [vp, vp, i, i, vp]),
    "nv_swiglu_bwd_bf16": (i, [vp, vp, vp, i, i, vp]),
    "nv_scale_bf16": (i, [vp, vp, l, f, vp]),
    "nv_scale_dev_bf16": (i, [vp, vp, l, fp, i, vp]),
    "nv_gather_rows_bf16": (i, [vp, ip, vp, i, i, vp]),
    "nv_scatter_rows_bf16": (i, [vp, ip, vp, i, i, vp]),
    "nv_attn_fwd_bf16": (i, [vp, vp, fp, ip, i, i, i, i, i, vp]),
    "nv_attn_fwd_strided_bf16": (i, [vp, vp, fp, ip, i, i, i, i, i, i, vp]),
    "nv_attn_decode_bf16": (i, [vp, ip, ip, vp, i, i, i, i, vp]),
    "nv_attn_fwd_strided_dyn_bf16": (i, [vp, vp, fp, ip, i, i, i, i, ip, vp]),
    "nv_attn_fwd_varlen_bf16": (i, [vp, vp, fp, ip, ip, i, i, i, i, i, vp]),
    "nv_attn_fwd_hfround_bf16": (i, [vp, vp, fp, ip, ip, i, i, i, i, i, vp]),
    "nv_attn_bwd_strided_bf16": (i, [vp, vp, vp, fp, ip, vp, vp, i, i, i, i, i, i, vp]),
    "nv_attn_bwd_strided_kvacc_bf16": (i, [vp, vp, vp, fp, ip, vp, vp, fp, ip, i, i, i, i, i, i, i, vp]),
    "nv_attn_fwd_episode_bf16": (i, [vp, vp, vp, ip, ip, i, i, i, i, i, i, l, vp]),
    "nv_attn_bwd_episode_bf16": (i, [vp, vp, vp, vp, vp, vp, ip, ip, fp, vp, vp, i, i, i, i, i, i, l, i, i, vp]),
    "nv_attn_bwd_episode_acc_bf16": (i, [vp, vp, vp, vp, vp, vp, ip, ip, fp, vp, vp, i, i, i, i, i, i, l, i, i, i, vp]),
    "nv_attn_bwd_varlen_bf16": (i, [vp, vp, vp, fp, ip, ip, vp, vp, vp, vp, i, i, l, i, i, i, vp]),
    "nv_attn_bwd_workspace_bytes": (sz, [i, i, i]),
    "nv_attn_bwd_bf16": (i, [vp, vp, vp, fp, ip, vp, vp, i, i, i, i, i, vp]),
    "nv_attn_bwd_rope_bf16": (i, [vp, vp, vp, fp, ip, vp, vp, vp, vp, i, i, i, i, i, vp]),
    "nv_head_fwd_bf16": (i, [vp, vp, vp, vp, i, i, i, vp]),
    "nv_head_bwd_bf16": (i, [vp, vp, vp, vp, vp, vp, i, i, i, vp]),
    "nv_action_ce_bf16": (i, [vp, lp, fp, vp, i, i, f, fp, vp]),
    "nv_lm_ce_bf16": (i, [vp, ip, fp, i, i, i, i, i, f, i, vp]),
    "nv_sumsq": (i, [vp, l, i, fp, C.POINTER(C.c_int), vp]),
    "nv_clip_coef": (i, [fp, i, f, fp, vp]),
    "nv_adamw": (i, [vp, vp, vp, vp, l, i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i, fp, vp]),
    "nv_adamw_zero_grad": (i, [vp, vp, vp, vp, l, i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i, fp, vp]),
    "nv_gemm_f32": (i, [i, fp, fp, fp, fp, i, i, i, i, i, i, i, vp]),
    "nv_gemm_f32_workspace_bytes": (sz, [i, i]),
    "nv_gemm_f32_ws": (i, [i, fp, fp, fp, fp, i, i, i, i, i, i, i, vp, vp]),
    "nv_layernorm_fwd_f32": (i, [fp, fp, fp, fp, fp, fp, i, i, f, vp]),
    "nv_layernorm_bwd_workspace_bytes": (sz, [i]),
    "nv_layernorm_bwd_f32": (i, [fp, fp, fp, fp, fp, fp, fp, fp, vp, i, i, i, vp]),
    "nv_colsum_f32": (i, [fp, fp, i, i, i, i, vp]),
    "nv_mha_fwd_f32": (i, [fp, ip, fp, fp, i, i, i, i, vp]),
    "nv_mha_bwd_f32": (i, [fp, fp, fp, fp, i, i, i, i, vp]),
    "nv_mha_fwd_drop_f32": (i, [fp, ip, fp, fp, fp, f, C.c_ulonglong, C.c_ulonglong, i, i, i, i, vp]),
    "nv_mha_bwd_drop_f32": (i, [fp, fp, fp, fp, fp, f, C.c_ulonglong, C.c_ulonglong, i, i, i, i, vp]),
    "nv_gelu_fwd_f32": (i, [fp, fp, l, vp]),
    "nv_gelu_bwd_f32": (i, [fp, fp, fp, l, vp]),
    "nv_add_f32": (i, [fp, fp, fp, l, i, i, vp]),
    "nv_mul_f32": (i, [fp, fp, fp, l, vp]),
    "nv_dropout_f32": (i, [fp, fp, l, f, C.c_ulonglong, C.c_ulonglong, vp]),
    "nv_rowscale_f32": (i, [fp, fp, fp, l, i, vp]),
    "nv_gather_add_f32": (i, [fp, ip, fp, fp, l, i, vp]),
    "nv_index_sum_f32": (i, [fp, ip, fp, i, i, i, i, vp]),
    "nv_masked_mean_f32": (i, [fp, fp, fp, i, i, i, vp]),
    # host side-car (nv_graph* travels as void*; all other pointers are HOST pointers)
    "nv_graph_create": (vp, []),
    "nv_graph_destroy": (None, [vp]),
    "nv_graph_add_node": (i, [vp]),
    "nv_graph_num_nodes": (i, [vp]),
    "nv_graph_set_position": (i, [vp, i, vp]),
    "nv_graph_add_edge": (i, [vp, i, i, C.c_double]),
    "nv_graph_update": (i, [vp, i]),
    "nv_graph_visited": (i, [vp, i]),
    "nv_graph_distance": (C.c_double, [vp, i, i]),
    "nv_graph_path": (i, [vp, i, i, vp, i]),
    "nv_graph_pos_fts": (i, [vp, i, vp, i, C.c_double, C.c_double, i, vp]),
    "nv_graph_set_step_id": (i, [vp, i, i]),
    "nv_nav_collate": (i, [vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nv_nav_match_tables": (i, [vp, vp, vp, i, i, i, vp, vp, vp]),
    "nv_nav_perm_tables": (i, [vp, vp, vp, i, i, vp, vp, vp]),
    # native inference runtime (nv_decoder* travels as void*)
    "nv_decoder_create": (vp, [i, i, i, i, i, f]),
    "nv_decoder_destroy": (None, [vp]),
    "nv_decoder_set_weight": (i, [vp, i, i, vp, vp, fp]),
    "nv_decoder_set_layer": (i, [vp, i, vp, vp, vp]),
    "nv_decoder_set_shared": (i, [vp, vp, vp, vp, vp, vp]),
    "nv_decoder_set_fp8_overlap": (i, [vp, vp, vp, vp]),
    "nv_decoder_set_fp8_gemm_mode": (i, [vp, i]),
    "nv_decoder_workspace_bytes": (sz, [vp, i]),
    "nv_decoder_extend": (i, [vp, vp, ip, ip, ip, ip, vp, fp, ip, vp, vp, i, i, i, i, i, ip, vp, sz, vp]),
    "nv_gemv_pre": (i, [vp, ip, vp, fp, vp, vp, i, i, i, i, i, i, i, i, vp, f, i, vp]),
    "nv_rope_scatter_rows_bf16": (i, [vp, vp, vp, ip, ip, vp, i, i, i, i, vp]),
    "nv_decode_state_ints": (i, [i]),
    "nv_decode_pick_bf16": (i, [vp, i, i, i, i, ip, ip, i, i, i, i, vp]),
    "nv_decode_advance": (i, [ip, i, i, vp]),
    "nv_decoder_greedy_step": (i, [vp, vp, vp, vp, i, i, i, i, vp, vp, ip, ip, i, ip, vp, fp, i, i, i, i, vp, sz, vp]),
    # data-parallel exchange over RCCL (nv_ctx* travels as void*)
    "nv_comm_unique_id_bytes": (i, []),
    "nv_comm_unique_id": (i, [vp]),
    "nv_comm_rccl_version": (i, []),
    "nv_comm_init": (i, [C.POINTER(C.c_void_p), vp, i, i]),
    "nv_comm_rank": (i, [vp]),
    "nv_comm_world": (i, [vp]),
    "nv_comm_allreduce_bf16": (i, [vp, vp, l, i, vp]),
    "nv_comm_allreduce_f32": (i, [vp, vp, l, i, vp]),
    "nv_comm_reduce_scatter": (i, [vp, vp, l, i, i, vp]),
    "nv_comm_all_gather": (i, [vp, vp, l, vp]),
    "nv_comm_broadcast": (i, [vp, vp, l, i, vp]),
    "nv_comm_destroy": (i, [vp]),
}

_lib = None


class NaviLLMHipError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (torch/lib, SONAME libamdhip64.so.7).  Import torch FIRST so that
    # our library's NEEDED libamdhip64.so.7 binds to that same runtime instance; loading ours first would
    # pull /opt/rocm's copy and leave two HIP runtimes in the process (torch's streams/pointers would then
    # be foreign handles to our launches).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NaviLLMHipError(
            f"{LIB_PATH} not found: build it with `python -m navillm_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NaviLLMHipError(f"libnavillm_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_ERR = {-1: "invalid argument", -2: "unsupported shape", -3: "launch failed"}


def check(rc, what):
    if rc != 0:
        raise NaviLLMHipError(f"{what} failed: {_ERR.get(rc, rc)}")
