"""GPU-only diagnostic: which ATen ops (i.e. non-HIP-kernel work) still run inside one native training step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel

dev = torch.device("cuda:0")
with torch.device("meta"):
    m = NativeUNet2DConditionModel()
m = m.to_empty(device=dev)
with torch.no_grad():
    for n, p in m.named_parameters():
        p.normal_(0, 0.02) if p.dim() > 1 else p.fill_(1.0 if "norm" in n and n.endswith("weight") else 0.0)
tr = NativeTrainer(m, [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=8)], lr=1e-4)
x = torch.randn(4, 4, 64, 64, device=dev); e = torch.randn(4, 77, 768, device=dev).to(torch.bfloat16)
for _ in range(2):
    tr.train_one_step(x, e)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    tr.train_one_step(x, e)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=70))
