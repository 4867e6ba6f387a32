"""GPU lab: product-library time of conv3x3 [B4,64,64,C] -> 320 and of the plain GEMM M16384 N320 K as a function of K: slope = cost of one
K tile (64 deep), intercept = everything outside the K loop (launch, prologue, epilogue)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
dev = torch.device("cuda:0")
def rnd(*s): return (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
rows = []
for C in (64, 128, 192, 256, 320, 640):
    x = rnd(4, 64, 64, C); w = rnd(320, 3, 3, C)
    t = timeit(lambda: K.conv3x3(x, w, 320))
    rows.append(("conv", 9 * C // 64, t))
    print(f"conv C{C:4d}->320 @64^2: K tiles {9 * C // 64:3d}  {t:6.1f} us", flush=True)
for Kd in (64, 320, 640, 1280, 2880, 5760):
    a, b = rnd(16384, Kd), rnd(320, Kd)
    t = timeit(lambda: K.gemm(a, b))
    print(f"gemm M16384 N320 K{Kd:5d}: K tiles {Kd // 64:3d}  {t:6.1f} us", flush=True)
for Kd in (64, 320, 1280, 2560):
    a, b = rnd(2048, Kd), rnd(1280, Kd)
    t = timeit(lambda: K.gemm(a, b))
    print(f"gemm M2048 N1280 K{Kd:5d}: K tiles {Kd // 64:3d}  {t:6.1f} us", flush=True)
