"""GPU lab: fused clip+AdamW over a full-fine-tune sized bucket (859.5 M fp32 parameters: 27.5 GB of traffic per launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
dev = torch.device("cuda:0")
n = 859_520_964 // 4 * 4
p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
g.fill_(1e-3)
lr = torch.tensor([1e-6], device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev); ss = torch.ones(1, device=dev)
for _ in range(3):
    K.adamw_clip_fused(p, g, m, v, lr, step, sumsq_t=ss, max_norm=1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.adamw_clip_fused(p, g, m, v, lr, step, sumsq_t=ss, max_norm=1.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"adamw over {n / 1e6:.1f} M parameters: {ms:.3f} ms = {8 * 4 * n / ms / 1e9:.2f} TB/s")
