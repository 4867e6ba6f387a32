"""Target of tools/pmc_passes.sh: conv C320 -> 320 at 64x64, B 4 (forward) through the ping-pong kernel and through conv_patch.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import _lib, kernels as K

K._set_backend_for_tests(_lib.load_tools())
dev = torch.device("cuda:0")
x = torch.randn(4, 64, 64, 320, device=dev).to(torch.bfloat16)
w = (torch.randn(320, 3, 3, 320, device=dev) * 0.02).to(torch.bfloat16)
for on in (0, 2):
    K.lib().hcp_debug_set_conv_patch(on)
    for _ in range(4):
        K.conv3x3(x, w, 320)
torch.cuda.synchronize()
