"""GPU diagnostic: per-tensor cosine of the native SDXL LoRA gradient against tests/golden/sdxl_full_b2_oracle.pt (which layers carry
the bf16-vs-fp32 difference of the full flat gradient?).  python tools/diag/sdxl_grad_diag.py"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.make_golden import sd15_lora_init_, sdxl_b2_inputs
from oracle.unet_sd15 import SDXL_CONFIG, seeded_init_

dev = torch.device("cuda:0")
g = torch.load(os.path.join(ROOT, "tests/golden/sdxl_full_b2_oracle.pt"))
with torch.device("meta"):
    nat = NativeUNet2DConditionModel(**SDXL_CONFIG)
nat = seeded_init_(nat.to_empty(device=dev), 1)
tr = NativeTrainer(nat, [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)], lr=1e-4)
by_name = {n: p for n, p in nat.named_parameters() if "lora_block_" in n}
lora_named = [(n, by_name[n]) for n in g["grad_names"]]
sd15_lora_init_(lora_named); tr.bucket.pack()
x0, ehs, noise, t, added = sdxl_b2_inputs()
added = {k: v.to(dev) for k, v in added.items()}
tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
with torch.no_grad():
    pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=added).sample.cpu()
print("pred rel-L2", ((pred - g["pred"].float()).norm() / g["pred"].float().norm()).item())
loss = tr.forward_backward(x0.to(dev), ehs.to(dev), None, added).item()
print("loss", loss, g["loss"])
off = 0
rows = []
for (n, p), s_ in zip(lora_named, g["grad_scales"].tolist()):
    k = p.numel()
    ref = g["grad_q"][off:off + k].double() * s_; off += k
    got = p.grad.detach().double().flatten().cpu()
    cos = float(ref @ got / (ref.norm() * got.norm() + 1e-300))
    rows.append((n, cos, float(ref.norm()), float(got.norm())))
groups = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for n, cos, rn, gn in rows:
    parts = n.split(".")
    blk = ".".join(parts[:2])
    kind = ("attn1" if ".attn1." in n else "attn2" if ".attn2." in n else "ff") + ("." + parts[-1])
    for key in (blk, kind, "ALL"):
        a = groups[key]; a[0] += cos * rn * gn; a[1] += rn * rn; a[2] += gn * gn; a[3] += 1
print("group  cosine  |ref|  |got|  n")
for key in sorted(groups):
    a = groups[key]
    print(f"{key:28s} {a[0] / (a[1] * a[2]) ** 0.5:.5f} {a[1] ** 0.5:.4f} {a[2] ** 0.5:.4f} {a[3]}")
rows.sort(key=lambda r: r[1])
print("worst 25:")
for n, cos, rn, gn in rows[:25]:
    print(f"  {cos:.4f} {rn:.3e} {gn:.3e} {n}")
import statistics
print("median per-tensor cosine", statistics.median(r[1] for r in rows))
