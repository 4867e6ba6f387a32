"""GPU lab: the 3x3 convolutions of the SD1.5 / SDXL steps with the LDS-resident input patch (csrc/conv_patch.hip) vs the ping-pong kernel
(hcp_debug_set_conv_patch), default dispatch, rotating operand sets (cold L2), tools library; then bench.py's step both ways.
   python tools/lab/conv_patch_ab.py                     per-shape table
   python tools/lab/conv_patch_ab.py step MODE <bench args>      (MODE: 0 never, 1 the product rule, 2 wherever eligible)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import _lib, kernels as K

K._set_backend_for_tests(_lib.load_tools())
L = K.lib()
BF = torch.bfloat16


def time_rot(calls, rounds=3):
    for c in calls:
        c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        for c in calls:
            c()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (rounds * len(calls))


def table():
    dev = torch.device("cuda:0")
    r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(BF)
    rows = []
    #      name, B, H, C1, C2, Cout, mode
    for name, B, H, C1, C2, Co, mode in [("fwd C320 64^2", 4, 64, 320, 0, 320, 0), ("fwd C640+320->320 64^2", 4, 64, 640, 320, 320, 0),
                                         ("fwd C320+320->320 64^2", 4, 64, 320, 320, 320, 0), ("dgrad C320 64^2", 4, 64, 320, 0, 320, 1),
                                         ("dgrad 320->960 64^2", 4, 64, 320, 0, 960, 1),
                                         ("fwd C640 32^2", 4, 32, 640, 0, 640, 0), ("fwd C1280+640->640 32^2", 4, 32, 1280, 640, 640, 0), ("dgrad C640 32^2", 4, 32, 640, 0, 640, 1),
                                         ("dgrad 640->1920 32^2", 4, 32, 640, 0, 1920, 1),
                                         ("fwd C1280 16^2", 4, 16, 1280, 0, 1280, 0), ("dgrad C1280 16^2", 4, 16, 1280, 0, 1280, 1), ("fwd C2560->1280 16^2", 4, 16, 1280, 1280, 1280, 0),
                                         ("sdxl fwd C640 64^2 b2", 2, 64, 640, 0, 640, 0), ("sdxl fwd C1280 32^2 b2", 2, 32, 1280, 0, 1280, 0), ("sdxl dgrad C1280 32^2 b2", 2, 32, 1280, 0, 1280, 1)]:
        nb = B * H * H * (C1 + C2 + Co) * 2 + Co * 9 * (C1 + C2) * 2
        nset = max(3, min(16, int(320e6 / nb)))
        sets = [(r(B, H, H, C1), r(B, H, H, C2) if C2 else None, r(Co, 3, 3, C1 + C2), r(B, H, H, Co)) for _ in range(nset)]
        if mode == 0:
            calls = [(lambda s=s: K.conv3x3(s[0], s[2], Co, x2=s[1], residual=s[3])) for s in sets]
        else:
            calls = [(lambda s=s: K.conv3x3(s[0], s[2], Co, mode=1, out_hw=(H, H))) for s in sets]
        t = {}
        for on in (0, 2, 0, 2):
            L.hcp_debug_set_conv_patch(on)
            t.setdefault(on, []).append(time_rot(calls))
        L.hcp_debug_set_conv_patch(1)
        a, b = min(t[0]), min(t[2])
        fl = 2.0 * B * H * H * Co * 9 * (C1 + C2)
        rows.append((name, a, b))
        print(f"{name:28s} ping-pong {a:8.1f} us   patch {b:8.1f} us   x{b / a:5.3f}   ({fl / b / 1e6:6.0f} TFLOP/s = {fl / b / 1e6 / 2500:.3f} of peak)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        L.hcp_debug_set_conv_patch(int(sys.argv[2]))
        sys.argv = [sys.argv[0]] + sys.argv[3:]
        import bench
        bench.main()
    else:
        table()
