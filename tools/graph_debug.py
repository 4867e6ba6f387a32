"""GPU-only diagnostic: hipGraph capture of the native step, stage by stage (run with python -X faulthandler)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel

TINY_CONFIG = dict(block_out_channels=(80, 160, 160, 160), layers_per_block=1, num_attention_heads=2, cross_attention_dim=64,
                   norm_num_groups=8)      # a small UNet with the SD1.5 head widths (40 / 80)

dev = torch.device("cuda:0")
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


def log(*a):
    print(*a, flush=True)


def stage_kernel_only():
    a = torch.randn(256, 128, device=dev).to(torch.bfloat16); b = torch.randn(64, 128, device=dev).to(torch.bfloat16)
    out = torch.empty(256, 64, device=dev, dtype=torch.bfloat16)
    K.gemm(a, b, out=out); torch.cuda.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        K.gemm(a, b, out=out)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        K.gemm(a, b, out=out)
    out.zero_(); g.replay(); torch.cuda.synchronize()
    log("stage1 single-kernel graph ok:", torch.equal(out, ref))


def build(cfg):
    with torch.device("meta"):
        m = NativeUNet2DConditionModel(**cfg)
    m = m.to_empty(device=dev)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1:
                p.normal_(0, p[0].numel() ** -0.5)
            elif "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    return m


def stage_forward(cfg, hw, ctx):
    m = build(cfg)
    m.requires_grad_(False)
    x = torch.randn(2, 4, hw, hw, device=dev); t = torch.tensor([10, 500], device=dev); e = torch.randn(2, 77, ctx, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        ref = m(x, t, e).sample.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = m(x, t, e).sample
        y.zero_(); g.replay(); torch.cuda.synchronize()
    log("stage2 forward graph ok:", torch.allclose(y, ref))


def stage_train(cfg, hw, ctx, steps=5):
    m = build(cfg)
    tr = NativeTrainer(m, [dict(layers=PATS, rank=4)], lr=1e-3, use_graph=True)
    x = torch.randn(2, 4, hw, hw, device=dev); e = torch.randn(2, 77, ctx, device=dev).to(torch.bfloat16)
    t0 = time.time()
    for i in range(steps):
        loss = tr.train_one_step(x, e)
        torch.cuda.synchronize()
        log(f"stage3 step {i} loss {loss.item():.5f} t={time.time() - t0:.2f}s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    stage_kernel_only()
    stage_forward(TINY_CONFIG, 16, 64)
    stage_train(TINY_CONFIG, 16, 64)
    if which == "full":
        stage_train({}, 64, 768, steps=5)
    log("done")
