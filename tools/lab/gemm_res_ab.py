"""GPU-only lab: GEMM / fused-LoRA GEMM / 3x3 convolution WITH a residual operand, two builds of the library interleaved in one process.
usage: python tools/lab/gemm_res_ab.py <other_build.so>     (the product build in the tree is "new", the argument is "old")
Rotating operand sets (cold L2), HIP events; checks that both builds return the same bits."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import _lib
from hcp_diffusion_amd import kernels as K

BF = torch.bfloat16
dev = torch.device("cuda:0")
new = _lib.load()
old = _lib.bind(ctypes.CDLL(os.path.abspath(sys.argv[1])))
NSET = 12


def timeit(fn, iters=4 * NSET, warm=NSET):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def r(*shape, s=1.0):
    return (torch.randn(*shape, device=dev) * s).to(BF)


cases = []
for (M, N, Kd) in [(16384, 320, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 1280, 5120), (2048, 1280, 1280), (8192, 640, 640)]:
    for res in (False, True):
        sets = [(r(M, Kd), r(N, Kd, s=0.05), r(32, Kd, s=0.05), r(N, 32, s=0.05), r(M, N)) for _ in range(NSET)]
        outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(NSET)]
        cases.append((f"gemm M{M} N{N} K{Kd}" + (" +res" if res else ""), lambda i, s=sets, o=outs, res=res: K.gemm(s[i][0], s[i][1], residual=s[i][4] if res else None, out=o[i])))
        cases.append((f"lora M{M} N{N} K{Kd}" + (" +res" if res else ""), lambda i, s=sets, res=res: K.gemm_lora(s[i][0], s[i][1], s[i][2], s[i][3], residual=s[i][4] if res else None)[0]))
for (B, H, C, Co) in [(4, 64, 320, 320), (4, 32, 640, 640), (4, 16, 1280, 1280)]:
    for res in (False, True):
        sets = [(r(B, H, H, C), r(Co, 3, 3, C, s=0.02), r(B * H * H, Co)) for _ in range(NSET)]
        cases.append((f"conv C{C}->{Co} @{H}x{H} B{B}" + (" +res" if res else ""), lambda i, s=sets, res=res, Co=Co: K.conv3x3(s[i][0], s[i][1], Co, residual=s[i][2] if res else None)))

for name, fn in cases:
    t = {"old": [], "new": []}
    outs = {}
    for rnd in range(2):
        for lname, lib in (("old", old), ("new", new)):
            K._set_backend_for_tests(lib)
            t[lname].append(timeit(lambda i: fn(i % NSET)))
            o = fn(0)
            outs[lname] = (o[0] if isinstance(o, tuple) else o).clone()
    same = torch.equal(outs["old"], outs["new"])
    to, tn = min(t["old"]), min(t["new"])
    print(f"{name:42s} old {to:7.2f} new {tn:7.2f} us ({tn / to - 1:+.1%})  same bits: {same}", flush=True)
