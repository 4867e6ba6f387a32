"""GPU-only: where does the implicit-GEMM conv main loop spend its time?  Ablates DMA / MFMA / LDS reads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library
from tune_gemm_common import timeit, rnd, CFG_NAMES

B, H, C = 4, 64, 320
x = rnd(B, H, H, C); w = rnd(C, 3, 3, C)
a, b = rnd(16384, 2880), rnd(320, 2880)
for cfg, split in [(7, 2), (7, 1), (3, 1), (8, 1), (4, 1), (6, 1)]:
    K.lib().hcp_debug_set_gemm_config(cfg + 16 * split)
    row = []
    for flags, tag in [(0, "full"), (1, "no-DMA"), (2, "no-MFMA"), (4, "no-LDSread/MFMA"), (3, "no-DMA,no-MFMA"), (5, "barriers only")]:
        K.lib().hcp_debug_set_gemm_ablation(flags)
        t_conv = timeit(lambda: K.conv3x3(x, w, C), iters=20)
        t_gemm = timeit(lambda: K.gemm(a, b), iters=20)
        row.append(f"{tag}: conv {t_conv:6.1f} gemm {t_gemm:6.1f}")
    K.lib().hcp_debug_set_gemm_ablation(0)
    print(f"{CFG_NAMES[cfg]:12s}/s{split} | " + " | ".join(row), flush=True)
K.lib().hcp_debug_set_gemm_config(-1)
