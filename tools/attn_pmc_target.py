"""Target for rocprofv3 --pmc: a few launches of each attention kernel at the SD1.5 64x64-level shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K

dev = torch.device("cuda:0")
B, H, N, D = 4, 8, 4096, 40
q, k, v, do = [torch.randn(B, N, H * D, device=dev).to(torch.bfloat16) for _ in range(4)]
for _ in range(3):
    o, lse = K.attention_fwd(q, k, v, H)
    K.attention_bwd(q, k, v, o, do, lse, H)
x = torch.randn(4, 64, 64, 320, device=dev).to(torch.bfloat16); w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(torch.bfloat16)
for _ in range(3):
    K.conv3x3(x, w, 320)
torch.cuda.synchronize()
