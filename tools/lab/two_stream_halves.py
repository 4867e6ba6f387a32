"""GPU lab: does running the step as TWO half-batches on two streams (parallel branches of one hipGraph) beat one full-batch chain?
The ~600 kernels of the step that sit at their launch floor leave most CUs idle; two independent chains could fill them.
(Timing only: the halves share split-K / GroupNorm workspaces here, so the numbers they produce are not checked.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd import kernels as K
dev = torch.device("cuda:0")
with torch.device("meta"):
    unet = NativeUNet2DConditionModel()
unet = unet.to_empty(device=dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for p in unet.parameters():
        p.normal_(0, 0.02, generator=g)
tr = NativeTrainer(unet, [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=8)], lr=1e-4, use_graph=False)
lat = torch.randn(4, 4, 64, 64, device=dev); ehs = torch.randn(4, 77, 768, device=dev).to(torch.bfloat16)
def timed(graph, n=20):
    for _ in range(3): graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): graph.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    tr.forward_backward(lat, ehs); tr.forward_backward(lat[:2].contiguous(), ehs[:2].contiguous())
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    K.wgrad_staging_begin_step(); tr.forward_backward(lat, ehs)
print(f"one chain, batch 4:              {timed(g1):7.3f} ms", flush=True)
l0, l1, e0, e1 = lat[:2].contiguous(), lat[2:].contiguous(), ehs[:2].contiguous(), ehs[2:].contiguous()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    K.wgrad_staging_begin_step(); tr.forward_backward(l0, e0); tr.forward_backward(l1, e1)
print(f"two chains of batch 2, in series: {timed(g2):7.3f} ms", flush=True)
for _ in range(4):                       # grow the descriptor staging pool past the slots the captures above froze
    tr.forward_backward(l0, e0)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    cur = torch.cuda.current_stream()
    K.wgrad_staging_begin_step()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): tr.forward_backward(l0, e0)
    with torch.cuda.stream(s2): tr.forward_backward(l1, e1)
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"two chains of batch 2, two streams: {timed(g3):7.3f} ms", flush=True)
