import ctypes
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.hookimpl(optionalhook=True)      # xdist's hook: absent under `-p no:xdist`
def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): workers for the CPU suite (interpreted kernels: CPU-bound, ~1 GB each), none for a run that selects the GPU
    tests (one device, full-size SD1.5 / SDXL models: they must not run side by side)."""
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    if "gpu" in expr and "notgpu" not in expr:
        return 0
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(6, n - 2 if n > 3 else n))


_emu_cdll = None


def emu_cdll():
    """Build (once) and open the TEST-ONLY CPU interpreter build of the kernels (tests/emu)."""
    global _emu_cdll
    if _emu_cdll is None:
        sys.path.insert(0, str(ROOT / "tests" / "emu"))
        from build_emu import build_emu
        # HCP_EMU_LIB: another build of the interpreter library — e.g. the AddressSanitizer build of tools/diag/emu_asan.sh, under which
        # every out-of-bounds global read / write of a kernel is a hard error (torch's CPU tensors are malloc'd: ASan red-zones them)
        _emu_cdll = ctypes.CDLL(os.environ.get("HCP_EMU_LIB") or str(build_emu()))
        # HCP_EMU_DMA_LATE=1: the whole run under the interpreter's second LDS-DMA timing model (copies land at the wait that retires them,
        # not at issue; tests/emu/hcp_emu.h) — `HCP_EMU_DMA_LATE=1 pytest -m "not gpu" tests/test_kernels.py` is the CPU race screen for
        # every counted-vmcnt protocol; the default run keeps the early-landing model and one dedicated late-landing test
        if os.environ.get("HCP_EMU_DMA_LATE") == "1":
            _emu_cdll.hcp_debug_emu_dma_deferred(1)
    return _emu_cdll


class Backend:
    def __init__(self, name):
        self.name = name
        self.device = torch.device("cuda:0" if name == "gpu" else "cpu")
        self.is_gpu = name == "gpu"

    def to(self, t):
        return t.to(self.device)


_gpu_checked = False


def gpu_box_check():
    """Once per GPU test session: print what box this is and run the exact fp32-atomics self-check of the product library
    (kernels.atomics_selfcheck) — a box whose atomic adds are wrong fails HERE, with its identity in the log, instead of in nine
    unrelated-looking tests (profiles/r5_gpu_tests_run_with_9_failures.txt)."""
    global _gpu_checked
    if _gpu_checked:
        return
    _gpu_checked = True
    from hcp_diffusion_amd import kernels
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        from box_info import box_info
        print(f"\n[box] {box_info()}")
    except Exception as e:  # noqa: BLE001 - identification is best effort
        print(f"\n[box] identification failed: {e}")
    prev = kernels._backend
    kernels._set_backend_for_tests(None)
    try:
        kernels.atomics_selfcheck("cuda:0")
        print("[box] fp32 atomics self-check: exact")
    finally:
        kernels._backend = prev


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """'emu': kernels interpreted on the CPU (tiny shapes, logic check);  'gpu': the gfx950 product library."""
    from hcp_diffusion_amd import kernels
    if request.param == "gpu":
        if not torch.cuda.is_available():
            pytest.skip("no GPU visible")
        gpu_box_check()
        kernels._set_backend_for_tests(None)      # product path: libhcp_mi355x.so, fails loudly if missing
        assert kernels.lib().hcp_is_emulated() == 0
    else:
        kernels._set_backend_for_tests(emu_cdll())
    yield Backend(request.param)
    kernels._set_backend_for_tests(None)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def tbackend(request):
    """Like `backend`, for the tests that force kernel variants through the hcp_debug_* hooks: on the GPU those exist only in the
    tuning build (libhcp_mi355x_tools.so: same kernel sources, -DHCP_TOOLS); the interpreter build always has them."""
    from hcp_diffusion_amd import _lib, kernels
    if request.param == "gpu":
        if not torch.cuda.is_available():
            pytest.skip("no GPU visible")
        gpu_box_check()
        kernels._set_backend_for_tests(_lib.load_tools())
        assert kernels.lib().hcp_is_emulated() == 0
    else:
        kernels._set_backend_for_tests(emu_cdll())
    yield Backend(request.param)
    kernels._set_backend_for_tests(None)
