"""GPU lab probe: prediction error of the native SDXL forward (configs[3] fixture inputs, LoRA rank 16 as in the fixture) against the fp32
oracle fixture, with the transformer blocks' residual stream in bf16 and as the (hi | lo) pair (unet.set_residual_stream; round 5 ran a torch-op fp32 prototype here) —
the reference's LoRA layers under autocast keep that stream in fp32 (mm + fp32 bias, lora_layers_patch.py:50-55); the calibration's
autocast figure is printed beside.  python tools/diag/sdxl_stream_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K, unet as U
from hcp_diffusion_amd.trainer import NativeTrainer
from oracle.make_golden import sd15_lora_init_, sdxl_b2_inputs
from oracle.unet_sd15 import SDXL_CONFIG, seeded_init_

dev = torch.device("cuda:0")
g = torch.load(os.path.join(ROOT, "tests/golden/sdxl_full_b2_oracle.pt"))
cal = torch.load(os.path.join(ROOT, "tests/golden/sdxl_b2_autocast_calibration.pt"))
with torch.device("meta"):
    nat = U.NativeUNet2DConditionModel(**SDXL_CONFIG)
nat = seeded_init_(nat.to_empty(device=dev), 1)
tr = NativeTrainer(nat, [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)], lr=1e-4)
by_name = {n: p for n, p in nat.named_parameters() if "lora_block_" in n}
sd15_lora_init_([(n, by_name[n]) for n in g["grad_names"]]); tr.bucket.pack()
x0, ehs, noise, t, added = sdxl_b2_inputs()
added = {k: v.to(dev) for k, v in added.items()}
ref = g["pred"].float()
for stream in (False, True):
    nat.set_residual_stream(stream)
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=added).sample.cpu()
    r = ((pred - ref).norm() / ref.norm()).item()
    print(f"residual stream {'(hi | lo) pair' if stream else 'bf16         '}: prediction rel-L2 {r:.4e}   ratio to autocast calibration ({cal['pred_rel']:.4e}): {r / cal['pred_rel']:.2f}", flush=True)
