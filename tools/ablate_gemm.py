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
a3, b3, l3, e3 = rnd(4096, 5120), rnd(640, 5120), rnd(32, 5120), rnd(640, 32)
V2 = [(0, "full"), (8, "no-DMA"), (16, "no-MFMA"), (32, "no-LDSread"), (48, "no-LDSread,no-MFMA"), (24, "no-DMA,no-MFMA"), (40, "no-DMA,no-LDSread"), (56, "barriers only")]
if len(sys.argv) > 1 and sys.argv[1] == "v2":
    # the DISPATCHED kernels (loader-wave v2 variants): where do conv C320@64^2, its plain-GEMM twin and a deep-K fused-LoRA linear spend time?
    for flags, tag in V2:
        K.lib().hcp_debug_set_gemm_ablation(flags)
        t_conv = timeit(lambda: K.conv3x3(x, w, C), iters=30)
        t_gemm = timeit(lambda: K.gemm(a, b), iters=30)
        t_lora = timeit(lambda: K.gemm_lora(a3, b3, l3, e3), iters=30)
        print(f"{tag:22s} conv C320@64^2 {t_conv:6.1f} us | gemm M16384 N320 K2880 {t_gemm:6.1f} us | fused-LoRA M4096 N640 K5120 {t_lora:6.1f} us", flush=True)
    K.lib().hcp_debug_set_gemm_ablation(0)
    sys.exit(0)
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
