"""GPU lab: true GPU time per launch (captured back-to-back in a hipGraph: no host gaps) of the 64x64-level projection shape with and without
the fused LoRA side path, and of the low-resolution shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
dev = torch.device("cuda:0")
def rnd(*s): return (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
def graph_time(fn, n=40):
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
for (M, N, Kd) in [(16384, 320, 320), (16384, 960, 320), (16384, 320, 1280), (4096, 640, 640), (1024, 1280, 1280), (1024, 1280, 5120), (256, 1280, 1280)]:
    a, b, l, e = rnd(M, Kd), rnd(N, Kd), rnd(32, Kd), rnd(N, 32)
    res = rnd(M, N); bias = torch.randn(N, device=dev)
    t_plain = graph_time(lambda: K.gemm(a, b))
    t_plain_r = graph_time(lambda: K.gemm(a, b, bias=bias, residual=res))
    t_lora = graph_time(lambda: K.gemm_lora(a, b, l, e))
    t_lora_r = graph_time(lambda: K.gemm_lora(a, b, l, e, bias=bias, residual=res))
    t_lora_not = graph_time(lambda: K.gemm_lora(a, b, l, e, want_t=False))
    fl = 2.0 * M * N * Kd
    print(f"M{M} N{N} K{Kd}: plain {t_plain:6.1f} us ({fl / t_plain / 1e6:5.0f} TF) | +bias+res {t_plain_r:6.1f} | fused LoRA {t_lora:6.1f} | LoRA+bias+res {t_lora_r:6.1f} | LoRA w/o T store {t_lora_not:6.1f}", flush=True)
