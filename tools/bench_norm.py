"""A/B of the GroupNorm kernels over the UNet's own shapes, straight through the C ABI with ctypes (HIP events on torch's stream).

  python tools/bench_norm.py [old_lib.so]

`old_lib.so` (optional): a library built from an earlier csrc/norm.hip (tools/build/, not tracked) — timed beside the product
library on identical buffers.  Also sweeps hcp_debug_set_gn_target.  Prints one line per (shape, variant): forward and
backward microseconds, and the time the bytes of each pass need at 6.3 TB/s for comparison."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hcp_diffusion_amd import _lib  # noqa: E402

SHAPES = [  # (B, HW, C, GroupNorms of that shape per SD1.5 forward) — resnet norm1/norm2 + transformer norms
    (4, 4096, 320, 13), (4, 4096, 640, 2), (4, 4096, 960, 1), (4, 1024, 320, 1), (4, 1024, 640, 11), (4, 1024, 960, 1),
    (4, 1024, 1280, 1), (4, 1024, 1920, 1), (4, 256, 640, 1), (4, 256, 1280, 12), (4, 256, 1920, 1), (4, 256, 2560, 2),
    (4, 64, 1280, 9), (4, 64, 2560, 2), (2, 16384, 320, 0), (2, 4096, 640, 0),
]


def bind(lib):
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.hcp_groupnorm_workspace_bytes.restype = ctypes.c_size_t
    lib.hcp_groupnorm_workspace_bytes.argtypes = [I, I, I, I]
    lib.hcp_groupnorm_silu_fwd.argtypes = [P, P, P, P, P, P, I, I, I, I, F, I, P]
    lib.hcp_groupnorm_silu_bwd.argtypes = [P, P, P, P, P, P, P, P, I, I, I, I, I, P]
    return lib


def time_us(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    new = bind(_lib.load())
    variants = [("new/512", new, 512), ("new/256", new, 256), ("new/1024", new, 1024)]
    if os.environ.get("GN_ONLY"):                    # for rocprofv3 --kernel-trace: one variant, per-kernel durations by shape order
        variants = [("new", new, int(os.environ["GN_ONLY"]))]
    elif len(sys.argv) > 1:
        variants.insert(0, ("old", bind(ctypes.CDLL(os.path.abspath(sys.argv[1]))), None))
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    totals = {v[0]: 0.0 for v in variants}
    for B, HW, C, count in SHAPES:
        x = (torch.randn(B, HW, C, device=dev) + 0.5).to(torch.bfloat16); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
        gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); stats = torch.empty(B, 32, 2, device=dev)
        ideal_f = 3 * x.numel() * 2 / 6.3e12 * 1e6
        ideal_b = 5 * x.numel() * 2 / 6.3e12 * 1e6
        ref = None
        for name, lib, target in variants:
            if target is not None:
                lib.hcp_debug_set_gn_target(target)
            ws = torch.empty(lib.hcp_groupnorm_workspace_bytes(B, HW, C, 32) // 4 + 16, device=dev)
            p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            f = lambda: lib.hcp_groupnorm_silu_fwd(p(x), p(gamma), p(beta), p(y), p(stats), p(ws), B, HW, C, 32, 1e-5, 1, stream)  # noqa: E731
            b = lambda: lib.hcp_groupnorm_silu_bwd(p(x), p(dy), p(gamma), p(beta), p(stats), None, p(dx), p(ws), B, HW, C, 32, 1, stream)  # noqa: E731
            assert f() == 0 and b() == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = (y.clone(), dx.clone())
            else:
                assert (y.float() - ref[0].float()).abs().max().item() < 0.05 and (dx.float() - ref[1].float()).abs().max().item() < 0.05
            tf, tb = time_us(f), time_us(b)
            totals[name] += count * (tf + tb)
            print(f"B{B} HW{HW:5d} C{C:4d} {name:9s} fwd {tf:6.1f} us (stream {ideal_f:5.1f})  bwd {tb:6.1f} us (stream {ideal_b:5.1f})", flush=True)
    new.hcp_debug_set_gn_target(512)
    print("per-step totals (SD1.5 bs4, us):", {k: round(v, 1) for k, v in totals.items()})


if __name__ == "__main__":
    main()
