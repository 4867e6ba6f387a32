"""VAE-encode throughput (images/s) of the native encoder at 512 px, random-init SD VAE weights: python tools/bench_vae.py [side] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hcp_diffusion_amd.vae import NativeVAEEncoder  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vae = NativeVAEEncoder().to(dev)
    for p in vae.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=(1.0 / (p.numel() / p.shape[0])) ** 0.5)
    for batch in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 4, 16]):
        img = torch.rand(batch, 3, side, side, device=dev) * 2 - 1
        for _ in range(2):
            vae.encode_latents(img)
        torch.cuda.synchronize()
        t0 = time.time(); n = 5
        for _ in range(n):
            vae.encode_latents(img)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        # encoder forward at 512 px ~ 566 GFLOP/image (convs 9*2*Cin*Cout*HW per layer + attention), for orientation only
        print(f"side {side} batch {batch}: {dt * 1e3:.1f} ms/batch, {batch / dt:.1f} images/s", flush=True)


if __name__ == "__main__":
    main()
