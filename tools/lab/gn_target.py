"""GPU lab (tools library): two-pass GroupNorm at the 64x64 level (C320 and C640 / C960 of the up blocks) for several workgroup targets;
every workgroup of the apply pass re-merges its sample's partials, so fewer, larger chunks trade merge work for parallelism."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())
dev = torch.device("cuda:0")
def graph_time(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3
for C in (320, 640, 960):
    x = (torch.randn(4, 64, 64, C, device=dev)).to(torch.bfloat16); dy = torch.randn_like(x)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    row = []
    for tgt in (128, 256, 384, 512, 768, 1024):
        K.lib().hcp_debug_set_gn_target(tgt)
        K._GN_WS.clear() if hasattr(K, "_GN_WS") else None
        y, st = K.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True)
        tf = graph_time(lambda: K.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True))
        tb = graph_time(lambda: K.groupnorm_bwd(x, dy, gamma, beta, st, 32, True))
        row.append(f"{tgt}: fwd {tf:5.1f} bwd {tb:5.1f}")
    print(f"C{C} @64^2 B4 | " + " | ".join(row), flush=True)
K.lib().hcp_debug_set_gn_target(512)
