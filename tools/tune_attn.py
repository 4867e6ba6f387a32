"""GPU-only: attention forward/backward timings per rows-per-wave configuration (hcp_debug_set_attention_config)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library

BF = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=3):
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


B, H = 4, 8
for (N, Nk, D) in [(4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (1024, 77, 80), (256, 256, 160), (256, 77, 160), (64, 64, 160)]:
    q, k, v, do = [torch.randn(B, n, H * D, device=dev).to(BF) for n in (N, Nk, Nk, N)]
    fl = 4.0 * B * H * N * Nk * D
    for cfg in ([0, 1, 2, 3, 4, 7] if D <= 64 else [0]):
        K.lib().hcp_debug_set_attention_config(cfg)
        tf = timeit(lambda: K.attention_fwd(q, k, v, H))
        o, lse = K.attention_fwd(q, k, v, H)
        tb = timeit(lambda: K.attention_bwd(q, k, v, o, do, lse, H))
        print(f"N{N} Nk{Nk} d{D} cfg{cfg}: fwd {tf:8.1f}us {fl / tf / 1e6:6.1f} TF | bwd {tb:8.1f}us {2.5 * fl / tb / 1e6:6.1f} TF", flush=True)
    K.lib().hcp_debug_set_attention_config(-1)
