"""GPU lab (tools library): conv C320@64^2 and a few GEMM shapes with the loader ring forced to 2 / 3 / 4 K tiles, GPU time from launches
captured back to back in a hipGraph.  Is a K tile's cost the DMA latency divided by the tiles in flight?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())
dev = torch.device("cuda:0")
def rnd(*s): return (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
def graph_time(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = rnd(4, 64, 64, 320); w = rnd(320, 3, 3, 320)
x6 = rnd(4, 32, 32, 640); w6 = rnd(640, 3, 3, 640)
a1, b1 = rnd(16384, 1280), rnd(320, 1280)
a2, b2, l2, e2 = rnd(1024, 1280), rnd(1280, 1280), rnd(32, 1280), rnd(1280, 32)
a3, b3 = rnd(4096, 2560), rnd(640, 2560)
cases = [("conv C320@64^2 (45 K tiles)", lambda: K.conv3x3(x, w, 320)), ("conv C640@32^2 (90 K tiles)", lambda: K.conv3x3(x6, w6, 640)),
         ("gemm M16384 N320 K1280", lambda: K.gemm(a1, b1)), ("fused-LoRA M1024 N1280 K1280", lambda: K.gemm_lora(a2, b2, l2, e2)),
         ("gemm M4096 N640 K2560", lambda: K.gemm(a3, b3))]
for name, fn in cases:
    row = []
    for ld in (-1, 1, 3, 4):
        K.lib().hcp_debug_set_gemm_loaders(ld)
        row.append(f"{'table' if ld < 0 else 'ring ' + str(2 if ld == 1 else ld)} {graph_time(fn):6.1f}")
    K.lib().hcp_debug_set_gemm_loaders(-1)
    print(f"{name:32s} " + " | ".join(row), flush=True)
