"""Micro-benchmark of the individual kernels at the SD1.5 bs=4 shapes (GPU only). Writes gpurun_out/kbench.json."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K

BF = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


res = []


def rec(name, t, flops=0, bytes_=0):
    r = {"name": name, "us": round(t * 1e6, 1), "TFLOPs": round(flops / t / 1e12, 1) if flops else None,
         "GBs": round(bytes_ / t / 1e9, 1) if bytes_ else None}
    print(r, flush=True)
    res.append(r)


def rnd(*s):
    return torch.randn(*s, device=dev).to(BF)


B = 4
for (M, N, Kd, K2) in [(16384, 320, 320, 32), (16384, 2560, 320, 32), (16384, 320, 1280, 32), (4096, 640, 640, 32), (4096, 5120, 640, 32),
                       (4096, 640, 2560, 32), (1024, 1280, 1280, 32), (1024, 10240, 1280, 32), (1024, 1280, 5120, 32), (256, 1280, 1280, 32),
                       (16384, 32, 320, 0), (16384, 32, 2560, 0), (308, 320, 768, 32), (8192, 8192, 8192, 0)]:
    a, b = rnd(M, Kd), rnd(N, Kd)
    a2, b2 = (rnd(M, K2), rnd(N, K2)) if K2 else (None, None)
    out = torch.empty(M, N, dtype=BF, device=dev)
    t = timeit(lambda: K.gemm(a, b, a2=a2, b2=b2, out=out))
    rec(f"gemm M{M} N{N} K{Kd}+{K2}", t, 2.0 * M * N * (Kd + K2), 2.0 * (M * Kd + N * Kd + M * N))

for (C1, C2, H, Cout, stride, up) in [(320, 0, 64, 320, 1, 0), (640, 0, 32, 640, 1, 0), (1280, 0, 16, 1280, 1, 0), (1280, 0, 8, 1280, 1, 0),
                                      (1280, 1280, 8, 1280, 1, 0), (640, 320, 64, 320, 1, 0), (320, 0, 64, 320, 2, 0), (640, 0, 32, 640, 1, 1),
                                      (8, 0, 64, 320, 1, 0), (320, 0, 64, 4, 1, 0)]:
    x1 = rnd(B, H, H, C1); x2 = rnd(B, H, H, C2) if C2 else None
    wp = rnd(Cout, 3, 3, C1 + C2)
    t = timeit(lambda: K.conv3x3(x1, wp, Cout, x2=x2, stride=stride, upsample=bool(up)))
    Ho = H * (2 if up else 1) // stride
    rec(f"conv3x3 C{C1}+{C2} H{H} Cout{Cout} s{stride} up{up}", t, 2.0 * B * Ho * Ho * Cout * 9 * (C1 + C2))
    if C2 == 0 and not up and C1 >= 320 and Cout >= 320:
        dy = rnd(B, Ho, Ho, Cout); wd = rnd(C1, 3, 3, Cout)
        t = timeit(lambda: K.conv3x3(dy, wd, C1, mode=1, stride=stride, out_hw=(H, H)))
        rec(f"dgrad3x3 C{C1} H{H} Cout{Cout} s{stride}", t, 2.0 * B * Ho * Ho * Cout * 9 * C1)

for (N, Nk, D) in [(4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (1024, 77, 80), (256, 256, 160), (256, 77, 160), (64, 64, 160)]:
    H = 8
    q, k, v, do = rnd(B, N, H * D), rnd(B, Nk, H * D), rnd(B, Nk, H * D), rnd(B, N, H * D)
    t = timeit(lambda: K.attention_fwd(q, k, v, H))
    fl = 4.0 * B * H * N * Nk * D
    rec(f"attn_fwd N{N} Nk{Nk} d{D}", t, fl)
    o, lse = K.attention_fwd(q, k, v, H)
    t = timeit(lambda: K.attention_bwd(q, k, v, o, do, lse, H))
    rec(f"attn_bwd N{N} Nk{Nk} d{D}", t, 2.5 * fl)

for (H, C) in [(64, 320), (64, 640), (32, 640), (32, 1280), (16, 1280), (16, 2560), (8, 1280), (8, 2560)]:
    x = rnd(B, H, H, C); g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev); dy = rnd(B, H, H, C)
    t = timeit(lambda: K.groupnorm_fwd(x, g, bt, 32, 1e-5, True))
    rec(f"gn_fwd H{H} C{C}", t, 0, 2.0 * x.numel() * 3)
    y, st = K.groupnorm_fwd(x, g, bt, 32, 1e-5, True)
    t = timeit(lambda: K.groupnorm_bwd(x, dy, g, bt, st, 32, True))
    rec(f"gn_bwd H{H} C{C}", t, 0, 2.0 * x.numel() * 5)
for (M, C) in [(16384, 320), (4096, 640), (1024, 1280)]:
    x = rnd(M, C); g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev); dy = rnd(M, C)
    t = timeit(lambda: K.layernorm_fwd(x, g, bt, 1e-5)); rec(f"ln_fwd M{M} C{C}", t, 0, 4.0 * M * C)
    y, st = K.layernorm_fwd(x, g, bt, 1e-5)
    t = timeit(lambda: K.layernorm_bwd(x, dy, g, st)); rec(f"ln_bwd M{M} C{C}", t, 0, 6.0 * M * C)
    h = rnd(M, 8 * C); dy4 = rnd(M, 4 * C)
    t = timeit(lambda: K.geglu_fwd(h)); rec(f"geglu_fwd M{M} F{4*C}", t, 0, 2.0 * M * 12 * C)
    t = timeit(lambda: K.geglu_bwd(h, dy4)); rec(f"geglu_bwd M{M} F{4*C}", t, 0, 2.0 * M * 20 * C)
for (M, Kd) in [(16384, 320), (16384, 2560), (4096, 5120)]:
    L = rnd(M, 32); R = rnd(M, Kd); out = torch.zeros(8, Kd, device=dev)
    t = timeit(lambda: K.lora_wgrad(L, R, out, 8, 1.0, False)); rec(f"lora_wgrad M{M} Q{Kd}", t, 2.0 * M * Kd * 32, 2.0 * M * (Kd + 32))

os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)
