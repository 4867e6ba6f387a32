"""GPU lab: the grouped LoRA weight-gradient launch (partial tiles + ordered reduce, csrc/lora.hip) on the layer list of the SD1.5 bs=4
headline step (160 layers, rank 8) and of SDXL bs=2 (700 layers, rank 16), for several grid targets (kernels.WGRAD_GRID_BLOCKS).
   python tools/lab/wgrad_grouped_bench.py [sd15|sdxl]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K

BF = torch.bfloat16
dev = torch.device("cuda:0")


def layers(kind):
    out = []
    if kind == "sd15":
        B, r, ctx = 4, 8, 768
        levels = [(4096, 320, 5), (1024, 640, 5), (256, 1280, 5), (64, 1280, 1)]
    else:
        B, r, ctx = 2, 16, 2048
        levels = [(4096, 640, 10), (1024, 1280, 60)]
    for hw, c, nblk in levels:
        M = B * hw
        for _ in range(nblk):
            out += [(M, c, c)] * 6 + [(B * 77, ctx, c)] * 2 + [(M, c, 8 * c), (M, 4 * c, c)]
    return out, r


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "sd15"
    shapes, r = layers(kind)
    pool = {}
    items, nbytes = [], 0
    for i, (M, Kd, N) in enumerate(shapes):
        # a few operand sets per shape, rotated: the step never re-reads a layer's operands from a warm cache
        key = (M, Kd, N, i % 3)
        if key not in pool:
            pool[key] = (torch.randn(M, 32, device=dev).to(BF), torch.randn(M, Kd, device=dev).to(BF), torch.randn(M, 32, device=dev).to(BF),
                         torch.randn(M, N, device=dev).to(BF))
        U, x, T, dy = pool[key]
        gd = torch.zeros(r, Kd, device=dev); gu = torch.zeros(N, r, device=dev)
        items.append((U, x, gd, T, dy, gu, r, 1.0))
        nbytes += 2 * M * (Kd + N + 64)
    print(f"{kind}: {len(items)} layers, {nbytes / 1e9:.2f} GB of operands per launch", flush=True)
    for target in (2048, 4096, 8192, 16384, 32768, 65536):
        K.WGRAD_GRID_BLOCKS = target
        keep = [K.lora_wgrad_grouped(items) for _ in range(2)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            keep.append(K.lora_wgrad_grouped(items))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"  grid target {target:6d}: {us:8.1f} us per grouped pass  ({nbytes / us / 1e6:.2f} TB/s of operand bytes)", flush=True)
    a = [t[2].clone() for t in items[:8]]
    for t in items:
        t[2].zero_(); t[5].zero_()
    keep.append(K.lora_wgrad_grouped(items)); b1 = torch.cat([t[2].flatten() for t in items] + [t[5].flatten() for t in items]).clone()
    for t in items:
        t[2].zero_(); t[5].zero_()
    keep.append(K.lora_wgrad_grouped(items)); b2 = torch.cat([t[2].flatten() for t in items] + [t[5].flatten() for t in items])
    print("  two passes bit-identical:", bool(torch.equal(b1, b2)), flush=True)


if __name__ == "__main__":
    main()
