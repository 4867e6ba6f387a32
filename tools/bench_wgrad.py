"""Time the host weight-gradient kernels (csrc/wgrad.hip) on SD1.5 layer shapes: heuristic vs forced (tile width, token splits).
   python tools/bench_wgrad.py [batch]"""
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


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    lin = []
    for hw, c in ((4096, 320), (1024, 640), (256, 1280)):
        M = B * hw
        lin += [(M, c, c), (M, 8 * c, c), (M, c, 4 * c)]
    lin += [(B * 77, 320, 768), (B * 77, 1280, 768), (B * 4096, 320, 640), (B * 64, 1280, 1280)]
    conv = [(320, 0, 64, 320, 1, 0), (640, 0, 32, 640, 1, 0), (1280, 0, 16, 1280, 1, 0), (1280, 0, 8, 1280, 1, 0), (1280, 1280, 8, 1280, 1, 0),
            (1280, 640, 32, 640, 1, 0), (640, 320, 64, 320, 1, 0), (320, 0, 64, 320, 2, 0), (1280, 0, 16, 1280, 1, 1), (8, 0, 64, 320, 1, 0)]
    cfgs = [0] + [wx + 256 * s for wx in (64, 128) for s in (1, 2, 4, 8, 16, 32, 64)]
    out = []
    for (M, N, Kd) in lin:
        dy = torch.randn(M, N, device=dev).to(BF); x = torch.randn(M, Kd, device=dev).to(BF); dw = torch.zeros(N, Kd, device=dev)
        row = {}
        for c in cfgs:
            K.lib().hcp_debug_set_wgrad_tile(c)
            row[c] = round(timeit(lambda: K.wgrad_linear(dy, x, dw)), 1)
        K.lib().hcp_debug_set_wgrad_tile(0)
        best = min((v, k) for k, v in row.items() if k)
        ideal = 2.0 * M * N * Kd / 2.5e15 * 1e6
        print(f"linear M={M} N={N} K={Kd}: heuristic {row[0]} us, best {best[0]} us @wx={best[1] & 255} split={best[1] >> 8}  (mfma-ideal {ideal:.1f} us)", flush=True)
        out.append(dict(kind="linear", M=M, N=N, K=Kd, times=row))
    for (C1, C2, H, Cout, stride, up) in conv:
        x1 = torch.randn(B, H, H, C1, device=dev).to(BF); x2 = torch.randn(B, H, H, C2, device=dev).to(BF) if C2 else None
        Ho = H * 2 if up else (H // stride)
        dy = torch.randn(B, Ho, Ho, Cout, device=dev).to(BF)
        cw = 4 if C1 == 8 else C1 + C2
        dw = torch.zeros(Cout, 3, 3, cw, device=dev)
        row = {}
        for c in cfgs:
            K.lib().hcp_debug_set_wgrad_tile(c)
            row[c] = round(timeit(lambda: K.wgrad_conv3x3(dy, x1, dw, x2=x2, stride=stride, upsample=bool(up))), 1)
        K.lib().hcp_debug_set_wgrad_tile(0)
        best = min((v, k) for k, v in row.items() if k)
        ideal = 2.0 * B * Ho * Ho * Cout * 9 * (C1 + C2) / 2.5e15 * 1e6
        print(f"conv C={C1}+{C2} H={H} Cout={Cout} s={stride} up={up}: heuristic {row[0]} us, best {best[0]} us @wx={best[1] & 255} "
              f"split={best[1] >> 8}  (mfma-ideal {ideal:.1f} us)", flush=True)
        out.append(dict(kind="conv", C1=C1, C2=C2, H=H, Cout=Cout, stride=stride, up=up, times=row))
    root = os.environ.get("GRAFT_REPO_ROOT", ".")
    json.dump(out, open(os.path.join(root, "gpurun_out", f"bench_wgrad_b{B}.json"), "w"))


if __name__ == "__main__":
    main()
