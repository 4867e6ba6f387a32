"""The C-ABI library loads and exports every symbol include/hcp_mi355x.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from hcp_diffusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "hcp_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hcp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_product_library_exports_every_declared_symbol():
    from hcp_diffusion_amd.build import build_product
    lib = ctypes.CDLL(str(build_product()))       # cross-compiles for gfx950 without a GPU
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} missing from libhcp_mi355x.so"
    lib.hcp_is_emulated.restype = ctypes.c_int
    assert lib.hcp_is_emulated() == 0 and lib.hcp_abi_version() == 1


def test_product_path_fails_loudly_without_gpu_tensors():
    """No CPU fallback: calling a kernel wrapper with host tensors on the product library raises."""
    import torch
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(None)
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.HcpError):
        K.gemm(a, a)


def test_argument_validation_returns_error_codes():
    lib = _lib.load()
    rc = lib.hcp_gemm_bf16(None, 8, None, 8, None, 8, 8, 8, 8, None, 0, None, 0, 0, None, None, 0, 1, None, 0, 1.0, 0, None, 0, None)
    assert rc < 0 and b"null" in lib.hcp_last_error()
    rc = lib.hcp_layernorm_fwd(None, None, None, None, None, 4, 7, 1e-5, None)
    assert rc < 0 and b"bad shape" in lib.hcp_last_error()
