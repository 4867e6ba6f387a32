"""ctypes binding of libhcp_mi355x.so (the C-ABI declared in include/hcp_mi355x.h).

The product path opens exactly one file — ``hcp_diffusion_amd/libhcp_mi355x.so`` built for gfx950 by
``hcp_diffusion_amd.build`` — and raises if it is missing: there is no CPU or PyTorch fallback.
(Tests may bind another shared object exporting the same ABI — the CPU interpreter build in
``tests/emu`` — through :func:`bind`; the package itself never does.)
"""
import ctypes
import os
from ctypes import c_int, c_long, c_float, c_void_p, c_size_t, c_char_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libhcp_mi355x.so"

P, I, L, F = c_void_p, c_int, c_long, c_float

_PROTOTYPES = {
    "hcp_last_error": (c_char_p, []),
    "hcp_is_emulated": (I, []),
    "hcp_abi_version": (I, []),
    # line, bucket, nb, stride, workgroups, stream
    "hcp_selfcheck_atomics": (I, [P, P, I, I, I, P]),
    # A, lda, B, ldb, D, ldd, M, N, K, A2, lda2, B2, ldb2, K2, bias, rowbias, rowbias_ld, rows_per_group,
    # residual, ldr, residual_lo, D_lo, gact, alpha, out_f32, workspace, workspace_bytes, stream
    "hcp_gemm_bf16": (I, [P, I, P, I, P, I, I, I, I, P, I, P, I, I, P, P, I, I, P, I, P, P, P, F, I, P, c_size_t, P]),
    # A, lda, B, ldb, L, E, Tout, ldt, D, ldd, M, N, K, bias, residual, ldr, residual_lo, D_lo, gact, workspace, workspace_bytes, stream
    "hcp_gemm_lora_bf16": (I, [P, I, P, I, P, P, P, I, P, I, I, I, I, P, P, I, P, P, P, P, c_size_t, P]),
    # A, lda, B, ldb, L, E, Tout, HG, DHG, M, F, K, workspace, workspace_bytes, stream
    "hcp_gemm_geglu_bwd_bf16": (I, [P, I, P, I, P, P, P, I, P, P, I, I, I, P, c_size_t, P]),
    "hcp_gemm_workspace_bytes": (c_size_t, [I, I]),
    # X1, C1, X2, C2, B, Hs, Ws, Ho, Wo, mode, stride, upsample, Wp, Cout, D, ldd, bias, rowbias, rowbias_ld,
    # residual, ldr, out_f32, workspace, workspace_bytes, stream
    "hcp_conv3x3_bf16": (I, [P, I, P, I, I, I, I, I, I, I, I, I, I, P, I, P, I, P, P, I, P, I, I, P, P, P, c_size_t, P]),
    # Q, K, V, O, lse, B, H, Nq, Nk, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale, stream
    "hcp_attention_fwd": (I, [P, P, P, P, P, I, I, I, I, I, L, I, L, I, L, I, L, I, F, P, L, I, P]),
    # Q, K, V, O, dO, lse, delta, dQ, dK, dV, B, H, Nq, Nk, D, strides..., scale, workspace, workspace_bytes, stream
    "hcp_attention_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, I, L, I, L, I, L, I, F, P, L, I, P, c_size_t, P]),
    "hcp_groupnorm_workspace_bytes": (c_size_t, [I, I, I, I]),
    # x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, stream
    "hcp_groupnorm_silu_fwd": (I, [P, P, P, P, P, P, I, I, I, I, F, I, P]),
    # x, dy, gamma, beta, stats, addend, dx, ws, B, HW, C, G, silu, stream
    "hcp_groupnorm_silu_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    # x, x_lo, gamma, beta, y, stats, M, C, eps, stream
    "hcp_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, P]),
    # x, x_lo, dy, gamma, stats, addend, addend_lo, dx, dx_lo, M, C, stream
    "hcp_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "hcp_geglu_fwd": (I, [P, P, L, I, P]),
    "hcp_geglu_bwd": (I, [P, P, P, L, I, P]),
    "hcp_add_bf16": (I, [P, P, P, L, P]),
    "hcp_copy2d_bf16": (I, [P, I, P, I, L, I, P]),
    "hcp_concat2_bf16": (I, [P, I, P, I, P, L, I, P]),
    "hcp_silu_fwd": (I, [P, P, L, P]),
    "hcp_silu_bwd": (I, [P, P, P, L, P]),
    "hcp_nchw_to_nhwc_bf16": (I, [P, I, P, I, I, I, I, P]),
    "hcp_nhwc_to_nchw_f32": (I, [P, P, I, I, I, I, P]),
    "hcp_upsample2x_bwd": (I, [P, P, I, I, I, I, P]),
    "hcp_timestep_embedding": (I, [P, P, I, I, F, P]),
    "hcp_timestep_embedding_f32": (I, [P, P, I, I, F, P]),
    # dY, ldy, X, ldx, dW, ldw, M, N, K, workspace, workspace_bytes, stream
    "hcp_wgrad_linear_bf16": (I, [P, I, P, I, P, I, I, I, I, P, c_size_t, P]),
    # dY, ldy, X1, C1, X2, C2, dW, Cw, B, Hs, Ws, Ho, Wo, Cout, stride, upsample, workspace, workspace_bytes, stream
    "hcp_wgrad_conv3x3_bf16": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I, I, I, P, c_size_t, P]),
    # Y, ldy, out, ldo, M, N, rows_per_group, stream
    "hcp_colsum_bf16": (I, [P, I, P, I, I, I, I, P]),
    # ema, p, n, step, inv_gamma, power, decay_max, stream
    "hcp_ema_update": (I, [P, P, L, P, F, F, F, P]),
    # src, dst_bf16, n, scale, zero_src, stream / src_bf16, dst, n, stream
    "hcp_cast_f32_bf16": (I, [P, P, L, F, I, P]),
    "hcp_cast_bf16_f32": (I, [P, P, L, P]),
    "hcp_pack_piece_bytes": (I, []),
    # pieces, count, total_tiles, stream
    "hcp_pack_weights": (I, [P, I, I, P]),
    # x, dy, gamma, beta, stats, dgamma, dbeta, B, HW, C, G, silu, stream
    "hcp_groupnorm_affine_grad": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    # x, dy, stats, dgamma, dbeta, M, C, stream
    "hcp_layernorm_affine_grad": (I, [P, P, P, P, P, I, I, P]),
    "hcp_add_noise": (I, [P, P, P, P, P, I, L, P]),
    "hcp_cfg_ddim_step": (I, [P, P, P, L, I, F, F, F, P]),
    "hcp_snr_loss_weight": (I, [P, P, P, I, I, F, P]),
    "hcp_quick_gelu": (I, [P, P, P, L, P]),
    "hcp_embedding_bf16": (I, [P, P, P, P, P, L, I, I, P]),
    "hcp_transpose_bf16": (I, [P, P, I, I, I, P]),
    "hcp_softmax_rows": (I, [P, L, P, L, I, I, F, P]),
    "hcp_vae_latent_sample": (I, [P, P, P, P, P, I, I, L, F, P]),
    "hcp_mse_masked_mean": (I, [P, P, P, I, P, P, P, I, I, I, F, P]),
    # L, ldl, l_lo, R, ldr, out, ldo, M, P, Q, scale, transpose_out, workspace, workspace_bytes, stream
    "hcp_lora_wgrad": (I, [P, I, I, P, I, P, I, I, I, I, F, I, P, c_size_t, P]),
    # U, ldu, x, ldx, K, grad_down, T, ldt, dY, ldy, N, grad_up, M, r, scale, workspace, workspace_bytes, stream
    "hcp_lora_wgrad_pair": (I, [P, I, P, I, I, P, P, I, P, I, I, P, I, I, F, P, c_size_t, P]),
    "hcp_split_hi_lo_bf16": (I, [P, P, c_long, I, P]),
    "hcp_lora_wgrad_group_geometry": (I, [I, I, I, I, I, P, P, P, P]),
    "hcp_lora_wgrad_group_desc_bytes": (I, []),
    # descs, count, total_blocks, total_tiles, slab_units, workspace, workspace_bytes, stream
    "hcp_lora_wgrad_grouped": (I, [P, I, I, I, c_long, P, c_size_t, P]),
    "hcp_lora_pack": (I, [P, I, P]),
    "hcp_lora_pack_desc_bytes": (I, []),
    "hcp_sumsq_f32": (I, [P, L, P, P]),
    # p, g, m, v, n, lr, beta1, beta2, eps, wd, sumsq, grad_scale, max_norm, step, stream
    "hcp_adamw_clip_fused": (I, [P, P, P, P, L, P, F, F, F, F, P, F, F, P, P]),
    # data-parallel exchange (RCCL, csrc/comm.hip)
    "hcp_comm_unique_id": (I, [P]),
    "hcp_comm_init": (I, [I, I, P, ctypes.POINTER(P)]),
    "hcp_comm_destroy": (I, [P]),
    "hcp_comm_rank": (I, [P]),
    "hcp_comm_world": (I, [P]),
    "hcp_allreduce_flat": (I, [P, P, c_size_t, I, P]),
    "hcp_reduce_scatter_flat": (I, [P, P, P, c_size_t, I, P]),
    "hcp_allgather_flat": (I, [P, P, P, c_size_t, I, P]),
}

# hcp_debug_* tuning hooks: exported by the -DHCP_TOOLS builds only (libhcp_mi355x_tools.so, tests/emu)
_TOOLS_PROTOTYPES = {
    "hcp_debug_set_gemm_config": (I, [I]),
    "hcp_debug_set_gemm_glds": (I, [I]),
    "hcp_debug_set_gemm_loaders": (I, [I]),
    "hcp_debug_set_gemm_epilogue": (I, [I]),
    "hcp_debug_set_conv_patch": (I, [I]),
    "hcp_debug_set_gn_target": (I, [I]),
    "hcp_debug_set_gemm_ablation": (I, [I]),
    "hcp_debug_set_attention_config": (I, [I]),
    "hcp_debug_set_wgrad_tile": (I, [I]),
    "hcp_debug_gemm_table_stats": (I, [P, P]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)
TOOLS_SYMBOLS = tuple(_TOOLS_PROTOTYPES)
TOOLS_LIB_PATH = Path(__file__).resolve().parent / "libhcp_mi355x_tools.so"


ABI_VERSION = 3          # include/hcp_mi355x.h HCP_ABI_VERSION: bumped whenever an exported signature or descriptor layout changes


class HcpError(RuntimeError):
    pass


def bind(cdll):
    """Attach argtypes/restype for every exported symbol; raises AttributeError if one is missing, HcpError when the library was
    built from another revision of the header (a stale .so would take shifted arguments silently).  The tuning hooks are
    bound when the library has them (tools / interpreter builds)."""
    cdll.hcp_abi_version.restype = I
    cdll.hcp_abi_version.argtypes = []
    got = cdll.hcp_abi_version()
    if got != ABI_VERSION:
        raise HcpError(f"{getattr(cdll, '_name', 'library')}: hcp_abi_version() = {got}, these bindings are for ABI {ABI_VERSION} "
                       "(rebuild with `python -m hcp_diffusion_amd.build`)")
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _TOOLS_PROTOTYPES.items():
        fn = getattr(cdll, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return cdll


_tools = None


def load_tools():
    """TOOLS / TESTS ONLY: the -DHCP_TOOLS build (hcp_debug_* hooks).  Pass the result to kernels._set_backend_for_tests()."""
    global _tools
    if _tools is None:
        import torch  # noqa: F401
        if not TOOLS_LIB_PATH.exists():
            raise HcpError(f"{TOOLS_LIB_PATH} not found: build it with `python -m hcp_diffusion_amd.build`")
        _tools = ctypes.CDLL(str(TOOLS_LIB_PATH))
        for name in _TOOLS_PROTOTYPES:
            getattr(_tools, name)                      # all hooks must be there
    return _tools


_lib = None


def load():
    """Return the bound product library, loading it on first use.  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  torch must map ITS libamdhip64 first: our .so then binds to the same HIP runtime
        if not LIB_PATH.exists():
            raise HcpError(
                f"{LIB_PATH} not found: build it with `python -m hcp_diffusion_amd.build` "
                "(hipcc, gfx950). hcp_diffusion_amd has no CPU/PyTorch fallback path.")
        _lib = bind(ctypes.CDLL(str(LIB_PATH)))
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().hcp_last_error() if _lib is not None else b""
        raise HcpError(f"{what} failed: {msg.decode() if msg else rc}")
