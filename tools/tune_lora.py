"""GPU-only: fused-LoRA GEMM (hcp_gemm_lora_bf16) per tile config vs the two-launch form (skinny T GEMM + K-extension GEMM)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library
from tune_gemm_common import timeit, rnd, CFG_NAMES

SHAPES = [(16384, 320, 320), (16384, 2560, 320), (16384, 320, 1280), (16384, 320, 2560), (16384, 1280, 320),
          (4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560), (4096, 640, 5120), (4096, 2560, 640),
          (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (1024, 1280, 10240), (1024, 5120, 1280),
          (256, 1280, 1280), (256, 10240, 1280), (256, 1280, 5120), (308, 320, 768), (308, 640, 768), (308, 1280, 768),
          # fused q|k|v and k|v projection groups (forward N = 3C / 2C, input-gradient K = 3C)
          (16384, 960, 320), (16384, 320, 960), (4096, 1920, 640), (4096, 640, 1920), (1024, 3840, 1280), (1024, 1280, 3840),
          (256, 3840, 1280), (256, 1280, 3840), (308, 2560, 768)]
out = []
for (M, N, Kd) in SHAPES:
    a, b, l, e = rnd(M, Kd), rnd(N, Kd), rnd(32, Kd), rnd(N, 32)
    K.lib().hcp_debug_set_gemm_config(-1)

    def two():
        t = K.gemm(a, l)
        return K.gemm(a, b, a2=t, b2=e)
    t_two = timeit(two)
    res = {}
    for cid in (0, 1, 2, 3, 4, 5, 6, 8, 9, 12, 13, 14, 15):
        K.lib().hcp_debug_set_gemm_config(cid + 16)
        res[CFG_NAMES[cid]] = round(timeit(lambda: K.gemm_lora(a, b, l, e)), 1)
    K.lib().hcp_debug_set_gemm_config(-1)
    best = min((v, k) for k, v in res.items())
    fl = 2.0 * M * N * (Kd + 32)
    print(f"lora M{M} N{N} K{Kd}: two-launch {t_two:7.1f}us | fused best {best[1]:8s} {best[0]:7.1f}us ({fl / best[0] / 1e6:6.1f} TF) | {res}", flush=True)
    out.append({"M": M, "N": N, "K": Kd, "two_launch_us": round(t_two, 1), "fused": res, "best": best[1], "best_us": best[0]})
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(root, "gpurun_out", "tune_lora.json"), "w"), indent=0)
