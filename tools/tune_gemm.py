"""GPU-only: time every tile configuration / split-K factor of the GEMM + implicit-conv kernel on the shapes the
SD1.5 bs=4 step actually launches. Writes gpurun_out/tune_gemm.json (consumed when fitting the dispatch heuristic)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library

BF = torch.bfloat16
dev = torch.device("cuda:0")
CFG_NAMES = ["128x128", "128x64", "64x64", "128x160", "64x160", "256x128", "256x160", "128x320", "128x160w8s3", "128x160w4s3", "256x160w8s3",
             "128x320w16", "256x160w16", "128x160w8", "64x160w8", "128x128w8"]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def rnd(*s):
    return torch.randn(*s, device=dev).to(BF)


def sweep(name, fn, flops, nk1, allow_split=True):
    res = {}
    for cid in range(len(CFG_NAMES)):
        for s in (1, 2, 4, 8, 16):
            if s > 1 and (not allow_split or nk1 // s < 4):
                continue
            K.lib().hcp_debug_set_gemm_config(cid + 16 * s)
            try:
                res[f"{CFG_NAMES[cid]}/s{s}"] = round(timeit(fn), 1)
            except Exception as e:  # noqa: BLE001
                res[f"{CFG_NAMES[cid]}/s{s}"] = None
    K.lib().hcp_debug_set_gemm_config(-1)
    heur = round(timeit(fn), 1)
    best = min((v, k) for k, v in res.items() if v)
    print(f"{name:50s} best {best[1]:12s} {best[0]:8.1f}us {flops / best[0] / 1e6:7.1f} TF | heuristic {heur:8.1f}us", flush=True)
    return {"name": name, "flops": flops, "best": best[1], "best_us": best[0], "heuristic_us": heur, "all": res}


out = []
B = 4
GEMMS = [(16384, 320, 320, 32), (16384, 320, 320, 0), (16384, 2560, 320, 32), (16384, 320, 1280, 32), (16384, 320, 2560, 32), (16384, 1280, 320, 32),
         (16384, 320, 960, 0), (16384, 320, 640, 0), (16384, 32, 320, 0), (16384, 32, 2560, 0), (16384, 32, 1280, 0),
         (4096, 640, 640, 32), (4096, 5120, 640, 32), (4096, 640, 2560, 32), (4096, 640, 5120, 32), (4096, 2560, 640, 32), (4096, 640, 1920, 0),
         (4096, 32, 640, 0), (4096, 32, 5120, 0),
         (1024, 1280, 1280, 32), (1024, 10240, 1280, 32), (1024, 1280, 5120, 32), (1024, 1280, 10240, 32), (1024, 5120, 1280, 32), (1024, 1280, 2560, 0),
         (1024, 32, 1280, 0), (256, 1280, 1280, 32), (256, 10240, 1280, 32), (256, 1280, 5120, 32), (308, 320, 768, 32), (308, 1280, 768, 32), (4, 1280, 1280, 0)]
for (M, N, Kd, K2) in GEMMS:
    a, b = rnd(M, Kd), rnd(N, Kd)
    a2, b2 = (rnd(M, K2), rnd(N, K2)) if K2 else (None, None)
    o = torch.empty(M, N, dtype=BF, device=dev)
    out.append(sweep(f"gemm M{M} N{N} K{Kd}+{K2}", lambda: K.gemm(a, b, a2=a2, b2=b2, out=o), 2.0 * M * N * (Kd + K2), Kd // 64, K2 == 0 or True))

CONVS = [(320, 0, 64, 320, 1, 0), (640, 320, 64, 320, 1, 0), (320, 320, 64, 320, 1, 0), (320, 0, 64, 320, 2, 0), (320, 0, 32, 640, 1, 0), (640, 0, 32, 640, 1, 0),
         (1280, 640, 32, 640, 1, 0), (640, 640, 32, 640, 1, 0), (640, 320, 32, 640, 1, 0), (640, 0, 32, 640, 2, 0), (640, 0, 32, 640, 1, 1), (640, 0, 16, 1280, 1, 0),
         (1280, 0, 16, 1280, 1, 0), (1280, 1280, 16, 1280, 1, 0), (1280, 640, 16, 1280, 1, 0), (1280, 0, 16, 1280, 2, 0), (1280, 0, 16, 1280, 1, 1),
         (1280, 0, 8, 1280, 1, 0), (1280, 1280, 8, 1280, 1, 0), (1280, 0, 8, 1280, 1, 1)]
for (C1, C2, H, Cout, stride, up) in CONVS:
    x1 = rnd(B, H, H, C1); x2 = rnd(B, H, H, C2) if C2 else None
    wp = rnd(Cout, 3, 3, C1 + C2)
    Ho = H * (2 if up else 1) // stride
    fl = 2.0 * B * Ho * Ho * Cout * 9 * (C1 + C2)
    out.append(sweep(f"conv C{C1}+{C2} H{H} Cout{Cout} s{stride} up{up}", lambda: K.conv3x3(x1, wp, Cout, x2=x2, stride=stride, upsample=bool(up)),
                     fl, 9 * (C1 + C2) // 64))
    if C2 == 0 and not up:
        dy = rnd(B, Ho, Ho, Cout); wd = rnd(C1, 3, 3, Cout)
        out.append(sweep(f"dgrad C{C1} H{H} Cout{Cout} s{stride}", lambda: K.conv3x3(dy, wd, C1, mode=1, stride=stride, out_hw=(H, H)), fl, 9 * Cout // 64))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tune_gemm.json", "w"), indent=0)
