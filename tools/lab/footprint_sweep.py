"""GPU-only lab: does a kernel's duration depend on how much DISTINCT memory the launches before it touched?
The same launch over N rotating operand sets, N = 6 ... 400 (footprint 0.1 ... 8 GB): Infinity-Cache-warm, HBM-cold, and beyond
(address-translation reach).  HIP events over 2 rounds of the N sets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K

BF = torch.bfloat16
dev = torch.device("cuda:0")


def r(*shape, s=1.0):
    return (torch.randn(*shape, device=dev) * s).to(BF)


def sweep(name, make, run, per_set_mb, counts):
    out = []
    for n in counts:
        sets = [make() for _ in range(n)]
        for i in range(n):
            run(sets[i])
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                run(sets[i])
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        out.append(f"{n} sets ({n * per_set_mb / 1024:.2f} GB): {best:.1f} us")
        del sets
        torch.cuda.empty_cache()
    print(f"{name}: " + " | ".join(out), flush=True)


counts = [6, 30, 100, 400]
M, N, Kd = 2048, 1280, 1280
sweep("lora M2048 N1280 K1280 +res", lambda: (r(M, Kd), r(N, Kd, s=0.05), r(32, Kd, s=0.05), r(N, 32, s=0.05), r(M, N), ),
      lambda s: K.gemm_lora(s[0], s[1], s[2], s[3], residual=s[4]), (M * Kd + N * Kd + 2 * M * N) * 2 / 2 ** 20, counts)
# same shape, weights shared by all sets (only the activations rotate): separates "weights from HBM" from "everything cold"
w = (r(N, Kd, s=0.05), r(32, Kd, s=0.05), r(N, 32, s=0.05))
sweep("lora M2048 N1280 K1280 +res, ONE weight set", lambda: (r(M, Kd), r(M, N)),
      lambda s: K.gemm_lora(s[0], w[0], w[1], w[2], residual=s[1]), (M * Kd + 2 * M * N) * 2 / 2 ** 20, counts)
M, N, Kd = 16384, 320, 1280
sweep("lora M16384 N320 K1280 +res", lambda: (r(M, Kd), r(N, Kd, s=0.05), r(32, Kd, s=0.05), r(N, 32, s=0.05), r(M, N)),
      lambda s: K.gemm_lora(s[0], s[1], s[2], s[3], residual=s[4]), (M * Kd + N * Kd + 2 * M * N) * 2 / 2 ** 20, [6, 30, 100])
sweep("conv C320->320 @64x64 B4", lambda: (r(4, 64, 64, 320), r(320, 3, 3, 320, s=0.02)),
      lambda s: K.conv3x3(s[0], s[1], 320), (2 * 4 * 64 * 64 * 320 + 9 * 320 * 320) * 2 / 2 ** 20, [6, 30, 100, 300])
g, b = torch.ones(640, device=dev), torch.zeros(640, device=dev)
sweep("layernorm M8192 C640", lambda: (r(8192, 640),), lambda s: K.layernorm_fwd(s[0], g, b, 1e-5), 2 * 8192 * 640 * 2 / 2 ** 20, [6, 100, 400])
