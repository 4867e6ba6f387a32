"""GPU-only yardstick: torch.matmul (hipBLASLt / rocBLAS) vs this repo's GEMM on a few SD1.5 shapes (plain NT GEMM, bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, Kd) in [(16384, 320, 2880), (16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 5760), (4096, 5120, 640),
                   (1024, 1280, 11520), (1024, 10240, 1280), (8192, 8192, 8192)]:
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_v = timeit(lambda: torch.matmul(a, b.t(), out=o))
    t_o = timeit(lambda: K.gemm(a, b, out=o))
    fl = 2.0 * M * N * Kd
    print(f"M{M} N{N} K{Kd}: vendor {t_v:8.1f} us {fl / t_v / 1e6:7.1f} TF | ours {t_o:8.1f} us {fl / t_o / 1e6:7.1f} TF", flush=True)
