"""GPU-only: latency ablation of the attention kernels (hcp_debug_set_attention_ablation): how much of each kernel's time is
waiting for the next tile's global loads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K

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
for (N, Nk, D) in [(4096, 4096, 40), (1024, 1024, 80), (4096, 77, 40)]:
    q, k, v, do = [torch.randn(B, n, H * D, device=dev).to(BF) for n in (N, Nk, Nk, N)]
    o, lse = K.attention_fwd(q, k, v, H)
    for flags in (0, 1, 2, 3):
        K.lib().hcp_debug_set_attention_ablation(flags)
        tf = timeit(lambda: K.attention_fwd(q, k, v, H))
        tb = timeit(lambda: K.attention_bwd(q, k, v, o, do, lse, H))
        print(f"N{N} Nk{Nk} d{D} ablation={flags}: fwd {tf:8.1f} us | bwd (delta+dq+dkv) {tb:8.1f} us", flush=True)
    K.lib().hcp_debug_set_attention_ablation(0)
