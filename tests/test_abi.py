"""The C-ABI library loads and exports every symbol include/hcp_mi355x.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from hcp_diffusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(tools=False):
    """Symbols a header declares: the product ABI (include/hcp_mi355x.h), or (tools=True) the hooks of include/hcp_mi355x_tools.h."""
    src = open(os.path.join(ROOT, "include", "hcp_mi355x_tools.h" if tools else "hcp_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hcp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.EXPORTED_SYMBOLS)
    assert header_symbols(tools=True) == sorted(_lib.TOOLS_SYMBOLS)


def test_product_library_exports_every_declared_symbol():
    from hcp_diffusion_amd.build import build_product
    lib = ctypes.CDLL(str(build_product()))       # cross-compiles for gfx950 without a GPU
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} missing from libhcp_mi355x.so"
    lib.hcp_is_emulated.restype = ctypes.c_int
    declared = int(re.search(r"#define HCP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "hcp_mi355x.h")).read()).group(1))
    assert lib.hcp_is_emulated() == 0 and lib.hcp_abi_version() == declared == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_revision():
    """A stale .so would take shifted arguments silently: bind() compares hcp_abi_version() with its own revision."""
    class Stale:
        _name = "stale.so"
        class hcp_abi_version:                      # noqa: N801 - stands in for a ctypes function pointer
            restype = None; argtypes = None
            def __new__(cls):
                return _lib.ABI_VERSION - 1
    with pytest.raises(_lib.HcpError, match="hcp_abi_version"):
        _lib.bind(Stale)


def test_product_library_has_no_tuning_hooks_but_the_tools_build_does():
    """No process-global mutable knob behind the product ABI: hcp_debug_* exist only in the -DHCP_TOOLS build."""
    import subprocess
    from hcp_diffusion_amd.build import build_product, build_tools
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if l.strip()}
    prod, tools = exported(build_product()), exported(build_tools())
    assert not [s for s in prod if s.startswith("hcp_debug_")]
    assert sorted(s for s in prod if s.startswith("hcp_")) == header_symbols()
    assert set(header_symbols(tools=True)) <= tools and set(header_symbols()) <= tools


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
    rc = lib.hcp_gemm_bf16(None, 8, None, 8, None, 8, 8, 8, 8, None, 0, None, 0, 0, None, None, 0, 1, None, 0, None, None, None, 1.0, 0, None, 0, None)
    assert rc < 0 and b"null" in lib.hcp_last_error()
    rc = lib.hcp_layernorm_fwd(None, None, None, None, None, None, 4, 7, 1e-5, None)
    assert rc < 0 and b"bad shape" in lib.hcp_last_error()


def test_every_entry_point_rejects_bad_arguments_without_launching():
    """Error convention of the boundary: negative return code + hcp_last_error(), never a crash.  Every check
    below fails in host-side validation, so no kernel is launched (works on the GPU-less build box)."""
    lib = _lib.load()
    N = None
    bad = {
        "hcp_conv3x3_bf16": (N, 8, N, 0, 1, 4, 4, 4, 4, 0, 1, 0, 1, N, 8, N, 8, N, N, 0, N, 0, 0, N, N, N, 0, N),
        "hcp_gemm_lora_bf16": (N, 8, N, 8, N, N, N, 32, N, 8, 8, 8, 8, N, N, 0, N, N, N, N, 0, N),
        "hcp_gemm_geglu_bwd_bf16": (N, 8, N, 8, N, N, N, 32, N, N, 8, 8, 8, N, 0, N),
        "hcp_attention_fwd": (N, N, N, N, N, 1, 1, 8, 8, 40, 0, 40, 0, 40, 0, 40, 0, 40, 0.1, N, 0, 0, N),
        "hcp_attention_bwd": (N, N, N, N, N, N, N, N, N, N, 1, 1, 8, 8, 40, 0, 40, 0, 40, 0, 40, 0, 40, 0.1, N, 0, 0, N, 0, N),
        "hcp_groupnorm_silu_fwd": (N, N, N, N, N, N, 1, 16, 30, 32, 1e-5, 1, N),          # C % G != 0
        "hcp_groupnorm_silu_bwd": (N, N, N, N, N, N, N, N, 1, 16, 32, 32, 1, N),
        "hcp_groupnorm_affine_grad": (N, N, N, N, N, N, N, 1, 16, 32, 32, 1, N),
        "hcp_layernorm_bwd": (N, N, N, N, N, N, N, N, N, 4, 8, N),
        "hcp_layernorm_affine_grad": (N, N, N, N, N, 4, 8, N),
        "hcp_wgrad_linear_bf16": (N, 8, N, 8, N, 8, 8, 8, 8, N, 0, N),
        "hcp_wgrad_conv3x3_bf16": (N, 8, N, 8, N, 0, N, 8, 1, 4, 4, 4, 4, 8, 1, 0, N, 0, N),
        "hcp_colsum_bf16": (N, 8, N, 8, 8, 8, 8, N),
        "hcp_pack_weights": (N, 0, 0, N),
        "hcp_lora_wgrad": (N, 8, 0, N, 8, N, 8, 8, 40, 8, 1.0, 0, N, 0, N),                     # P > 32
        "hcp_lora_wgrad_pair": (N, 48, N, 8, 8, N, N, 32, N, 8, 8, N, 8, 8, 1.0, N, 0, N),
        "hcp_lora_wgrad_grouped": (N, 1, 1, 1, 1, N, 0, N),                                    # null table / no workspace
        "hcp_sumsq_f32": (N, 0, N, N),
        "hcp_adamw_clip_fused": (N, N, N, N, 0, N, 0.9, 0.999, 1e-8, 0.0, N, 1.0, 1.0, N, N),
        "hcp_ema_update": (N, N, 0, N, 1.0, 0.6, 0.99, N),
        "hcp_cast_f32_bf16": (N, N, 0, 1.0, 0, N),
        "hcp_cast_bf16_f32": (N, N, 0, N),
        "hcp_timestep_embedding": (N, N, 0, 3, 1e4, N),
        "hcp_add_noise": (N, N, N, N, N, 0, 0, N),
        "hcp_cfg_ddim_step": (N, N, N, 0, 1, 7.5, 0.5, 0.6, N),
        "hcp_snr_loss_weight": (N, N, N, 0, 0, 5.0, N),
        "hcp_quick_gelu": (N, N, N, 7, N),
        "hcp_embedding_bf16": (N, N, N, N, N, 0, 8, 77, N),
        "hcp_transpose_bf16": (N, N, 0, 8, 8, N),
        "hcp_softmax_rows": (N, 8, N, 8, 0, 8, 1.0, N),
        "hcp_vae_latent_sample": (N, N, N, N, N, 1, 9, 16, 1.0, N),
        "hcp_mse_masked_mean": (N, N, N, 1, N, N, N, 0, 0, 0, 1.0, N),
        "hcp_copy2d_bf16": (N, 8, N, 8, 0, 7, N),
        "hcp_concat2_bf16": (N, 8, N, 8, N, 4, 0, N),
    }
    for name, args in bad.items():
        assert name in _lib.EXPORTED_SYMBOLS, name
        rc = getattr(lib, name)(*args)
        assert rc < 0 and len(lib.hcp_last_error()) > 0, name
    # unsupported head dimension is reported, not mis-dispatched
    one = ctypes.c_void_p(16)
    rc = lib.hcp_attention_fwd(one, one, one, one, one, 1, 1, 8, 8, 48, 0, 48, 0, 48, 0, 48, 0, 48, 0.1, None, 0, 0, None)
    assert rc < 0 and b"head_dim" in lib.hcp_last_error()
    rc = lib.hcp_attention_fwd(one, one, one, one, one, 1, 1, 8, 16, 40, 0, 40, 0, 40, 0, 40, 0, 40, 0.1, None, 0, 1, None)   # causal cross-attention
    assert rc < 0 and b"causal" in lib.hcp_last_error()
    rc = lib.hcp_attention_fwd(one, one, one, one, one, 1, 1, 8, 8, 48, 0, 48, 0, 48, 0, 48, 0, 48, 0.1, None, 0, 0, None)
    assert rc < 0 and b"head_dim" in lib.hcp_last_error()


def test_integration_doc_quotes_the_shipped_overlay():
    """INTEGRATION.md shows the seam-1 YAML a maintainer copies: it must BE the shipped file, not a paraphrase of it."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    doc = (root / "INTEGRATION.md").read_text()
    for name in ("lora_sd15_hip.yaml", "lora_sdxl_hip.yaml"):
        assert (root / "cfgs" / "train" / "mi355x" / name).read_text() in doc, name


def test_integration_doc_ctypes_stub_matches_the_binding():
    """INTEGRATION.md section 3 shows a ctypes stub for hcp_gemm_bf16: its argtypes list and ABI revision must be the binding's own."""
    import pathlib
    doc = (pathlib.Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    m = re.search(r"lib\.hcp_gemm_bf16\.argtypes = \[(.*?)\]", doc)
    names = {"P": ctypes.c_void_p, "I": ctypes.c_int, "F": ctypes.c_float, "ctypes.c_size_t": ctypes.c_size_t}
    got = [names[t.strip()] for t in m.group(1).split(",")]
    assert got == list(_lib._PROTOTYPES["hcp_gemm_bf16"][1])
    assert f"lib.hcp_abi_version() == {_lib.ABI_VERSION}" in doc


def test_every_overlay_parses_and_extends_a_reference_example():
    """cfgs/train/mi355x/*.yaml: valid YAML, one `_base_` that names an example the reference ships (checked against /root/reference where
    that tree exists), bf16, and every `_target_` under hcp_diffusion_amd resolves to a real attribute."""
    import importlib
    import pathlib
    import yaml
    root = pathlib.Path(__file__).resolve().parent.parent
    files = sorted((root / "cfgs" / "train" / "mi355x").glob("*.yaml"))
    assert {f.name for f in files} >= {"lora_sd15_hip.yaml", "lora_sdxl_hip.yaml", "lora_sd15_te_hip.yaml", "dreambooth_sd15_hip.yaml", "controlnet_sd15_hip.yaml"}

    def targets(node):
        if isinstance(node, dict):
            for k, v in node.items():
                if k == "_target_":
                    yield v
                else:
                    yield from targets(v)
        elif isinstance(node, list):
            for v in node:
                yield from targets(v)
    for f in files:
        cfg = yaml.safe_load(f.read_text())
        assert len(cfg["_base_"]) >= 1, f.name
        for base in cfg["_base_"]:
            if base.startswith("cfgs/train/mi355x/"):                    # an overlay on a sibling overlay inherits its settings
                assert (root / base).exists(), (f.name, base)
                continue
            assert cfg["mixed_precision"] == "bf16", f.name
            if os.path.isdir("/root/reference"):
                assert os.path.exists(os.path.join("/root/reference", base)), (f.name, base)
        for t in targets(cfg):
            if t.startswith("hcp_diffusion_amd."):
                mod, _, attr = t.rpartition(".")
                obj = None
                while mod:
                    try:
                        obj = importlib.import_module(mod); break
                    except ModuleNotFoundError:
                        mod, _, head = mod.rpartition("."); attr = head + "." + attr
                for part in attr.split("."):
                    obj = getattr(obj, part)


def test_no_compiler_vmcnt_wait_drains_the_attention_tile_prefetch(tmp_path):
    """ISA hygiene (found in round 4 in the dK/dV kernel): the LDS-DMA tile fills are inline asm, invisible to hipcc's waitcnt pass, but the
    hardware counter is shared — a compiler-inserted `s_waitcnt vmcnt(N)` for some VGPR-destination load placed behind the DMA issue of a
    loop iteration, with MFMA work still behind it, makes the wave sit out the whole L2 -> LDS latency every tile.  The unmasked kernels
    (the ones every UNet step runs) must have none; tools/diag/loop_vmcnt_scan.py is the scanner."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = tmp_path / "attention.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(root, "hcp_diffusion_amd", "csrc"),
                    os.path.join(root, "hcp_diffusion_amd", "csrc", "attention.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "loop_vmcnt_scan.py"), str(asm)], capture_output=True, text=True)
    flagged = [l for l in r.stdout.splitlines() if l.startswith("hcp_attn::")]
    # masked / causal instantiations (KB = true) load the key bias per tile by design; <64, 2, false, 67> (non-pre-scaled d = 64 at 32 rows per
    # wave: not on any training path) reloads one spilled DMA offset
    bad = [l for l in flagged if ", true, 67" not in l and "attn2_bwd_dq_kernel<64, 2, false, 67>" not in l]
    assert not bad, "\n".join(bad)
    assert any("dkv" in l or "dq" in l or "fwd" in l for l in flagged) or not flagged
