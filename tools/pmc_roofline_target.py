"""Target of tools/pmc_passes.sh for bench.py's roofline kernels: a few launches of the conv3x3 C320->320 @64x64 B4 implicit GEMM
and of the attention forward / backward at B4 H8 N4096 d40 (summarise with tools/pmc_roofline.py -> profiles/pmc_roofline.json)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K

dev = torch.device("cuda:0")
x = torch.randn(4, 64, 64, 320, device=dev).to(torch.bfloat16)
w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(torch.bfloat16)
q, k, v, do = [torch.randn(4, 4096, 320, device=dev).to(torch.bfloat16) for _ in range(4)]
q = (q.float() * (40 ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)      # the form the UNet calls the kernels in: pre-scaled Q
for _ in range(4):
    K.conv3x3(x, w, 320)
    o, lse = K.attention_fwd(q, k, v, 8, q_prescaled=True)
    K.attention_bwd(q, k, v, o, do, lse, 8, q_prescaled=True)
torch.cuda.synchronize()
