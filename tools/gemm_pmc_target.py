"""Target for rocprofv3 --pmc: the deep-K plain GEMM (M16384 N320 K2880 = the conv C320@64x64 GEMM shape) with the 256x160 16-wave tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library

dev = torch.device("cuda:0")
a = torch.randn(16384, 2880, device=dev).to(torch.bfloat16); b = torch.randn(320, 2880, device=dev).to(torch.bfloat16)
o = torch.empty(16384, 320, dtype=torch.bfloat16, device=dev)
K.lib().hcp_debug_set_gemm_config(12 + 16 * 1)          # 256x160, 16 waves, no split: 128 workgroups... use the unsplit form to read pure main-loop behaviour
for _ in range(3):
    K.gemm(a, b, out=o)
K.lib().hcp_debug_set_gemm_config(13 + 16 * 1)          # 128x160, 8 waves (4x2), no split: 256 workgroups
for _ in range(3):
    K.gemm(a, b, out=o)
torch.cuda.synchronize()
