"""GPU-only: cross-attention backward (77 keys) vs the query-loop split of the dK/dV kernel (min query tiles per workgroup)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hcp_diffusion_amd import kernels as K, _lib
K._set_backend_for_tests(_lib.load_tools())
from tune_gemm_common import timeit, rnd

for (B, H, Nq, Nk, D) in [(4, 8, 4096, 77, 40), (4, 8, 1024, 77, 80), (4, 8, 256, 77, 160), (4, 8, 64, 77, 160), (2, 10, 4096, 77, 64)]:
    C = H * D
    q, do = rnd(B, Nq, C), rnd(B, Nq, C)
    k, v = rnd(B, Nk, C), rnd(B, Nk, C)
    o, lse = K.attention_fwd(q, k, v, H)
    row = []
    for mt, tg in ((8, 2), (16, 1), (32, 1), (64, 1), (128, 1)):
        K.lib().hcp_debug_set_attention_config(16 + (mt << 8) + (tg << 16))
        row.append(f"min {mt} target {256 * tg}: {timeit(lambda: K.attention_bwd(q, k, v, o, do, lse, H), iters=20):6.1f} us")
    K.lib().hcp_debug_set_attention_config(-1)
    print(f"B{B} H{H} Nq{Nq} Nk{Nk} d{D} backward (delta + dQ + memset + dK/dV + convert): " + " | ".join(row), flush=True)
