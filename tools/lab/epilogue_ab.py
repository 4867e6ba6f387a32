"""GPU lab: lane-layout epilogue (8-byte pieces in the MFMA layout) vs the tile epilogue (fp32 through LDS, 16-byte row pieces) of the GEMM
family, per shape of the SD1.5 / SDXL steps, on rotating operand sets (cold L2), tools library.
   python tools/lab/epilogue_ab.py            per-shape table
   python tools/lab/epilogue_ab.py step MODE <bench args>     bench.py's main() on the tools library with hcp_debug_set_gemm_epilogue(MODE)"""
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
    cases = []
    for name, M, N, Kd, lora, res in [("ff.proj 64^2 (LoRA)", 16384, 2560, 320, True, False), ("qkv 64^2 (LoRA)", 16384, 960, 320, True, False),
                                      ("to_out 64^2 (LoRA + res)", 16384, 320, 320, True, True), ("ff.out 64^2 (LoRA + res)", 16384, 320, 1280, True, True),
                                      ("ff.out dX 64^2 plain", 16384, 1280, 320, False, False), ("ff.proj dX 64^2", 16384, 320, 2560, False, False),
                                      ("ff.proj 32^2 (LoRA)", 4096, 5120, 640, True, False), ("to_out 32^2 (LoRA + res)", 4096, 640, 640, True, True),
                                      ("ff.proj 16^2 (LoRA)", 1024, 10240, 1280, True, False), ("to_out 16^2 (LoRA + res)", 1024, 1280, 1280, True, True),
                                      ("sdxl ff.proj (LoRA)", 2048, 10240, 1280, True, False), ("sdxl to_out (LoRA + res)", 2048, 1280, 1280, True, True),
                                      ("sdxl ff.proj 64^2 (LoRA)", 8192, 5120, 640, True, False)]:
        nset = max(3, min(24, int(300e6 / (2 * (M * Kd + N * Kd + M * N * (2 if res else 1))))))
        sets = [(r(M, Kd), r(N, Kd), r(32, Kd), r(N, 32), r(M, N) if res else None) for _ in range(nset)]
        if lora:
            calls = [(lambda s=s: K.gemm_lora(s[0], s[1], s[2], s[3], residual=s[4])) for s in sets]
        else:
            calls = [(lambda s=s: K.gemm(s[0], s[1], residual=s[4])) for s in sets]
        cases.append((name, calls, 2.0 * M * N * Kd, 2.0 * M * N))
    B, H, C = 4, 64, 320
    csets = [(r(B, H, H, C), r(C, 3, 3, C), r(B, H, H, C)) for _ in range(16)]
    cases.append(("conv C320 64^2 + res", [(lambda s=s: K.conv3x3(s[0], s[1], C, residual=s[2])) for s in csets], 2.0 * B * H * H * C * 9 * C, 2.0 * B * H * H * C))
    csets2 = [(r(B, 32, 32, 640), r(640, 3, 3, 640), r(B, 32, 32, 640)) for _ in range(16)]
    cases.append(("conv C640 32^2 + res", [(lambda s=s: K.conv3x3(s[0], s[1], 640, residual=s[2])) for s in csets2], 2.0 * B * 32 * 32 * 640 * 9 * 640, 2.0 * B * 32 * 32 * 640))
    print(f"{'shape':30s} {'lane us':>9s} {'tile us':>9s} {'tile/lane':>9s}   output MB")
    for name, calls, flops, obytes in cases:
        t = {}
        for mode in (0, 1, 0, 1):
            L.hcp_debug_set_gemm_epilogue(mode)
            t.setdefault(mode, []).append(time_rot(calls))
        L.hcp_debug_set_gemm_epilogue(-1)
        a, b = min(t[0]), min(t[1])
        print(f"{name:30s} {a:9.1f} {b:9.1f} {b / a:9.3f}   {obytes / 1e6:.1f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        L.hcp_debug_set_gemm_epilogue(int(sys.argv[2]))
        sys.argv = [sys.argv[0]] + sys.argv[3:]
        import bench
        bench.main()
    else:
        table()
