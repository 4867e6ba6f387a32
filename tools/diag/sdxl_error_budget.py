"""CPU diagnostic (round 6): where does the reference's LoRA-under-autocast mode get its precision from?
The reference's LoRA linears compute mm(x, W^T) [bf16 under autocast] + bias [fp32] = an fp32 output (lora_layers_patch.py:50-57): every
BIASED LoRA'd linear (to_out.0, ff.net.0.proj, ff.net.2) hands fp32 on.  This script runs a small SDXL-shaped oracle UNet with LoRA under
torch.autocast(bfloat16) and switches that promotion off per layer CLASS, against the same model in fp32:
   A  reference mode           : all three classes return fp32
   B  every output bf16        : what a bf16-stream implementation (this repo's default) computes
   C  stream writers fp32      : to_out.0 and ff.net.2 fp32 (the transformer blocks' residual stream, forward and backward), ff.net.0.proj bf16
   D  GEGLU input fp32 only    : ff.net.0.proj fp32, the stream bf16
   python tools/diag/sdxl_error_budget.py [seeds]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle.lora_ref import OracleLoraLinear, wrap_lora
from oracle.make_golden import sd15_lora_init_
from oracle.unet_sd15 import TINY_SDXL_CONFIG, OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod, seeded_init_

cfg = dict(TINY_SDXL_CONFIG, transformer_layers_per_block=(1, 2, 6))
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
_fwd = OracleLoraLinear.forward


def build(seed):
    m = seeded_init_(OracleUNet2DConditionModel(**cfg), seed)
    m.requires_grad_(False)
    wr = wrap_lora(m, PATS, rank=16)
    sd15_lora_init_([(n, p) for n, p in m.named_parameters() if "lora_block_" in n])
    for path, w in wr.items():
        w._path = path
    return m, wr


def klass(path):
    if path.endswith("to_out.0"):
        return "to_out"
    if path.endswith("ff.net.0.proj"):
        return "ff_proj"
    if path.endswith("ff.net.2"):
        return "ff_out"
    return "qkv"


def set_mode(fp32_classes):
    def fwd(self, x):
        y = _fwd(self, x)
        if torch.is_autocast_enabled("cpu") and klass(self._path) not in fp32_classes:
            return y.to(torch.bfloat16)
        return y
    OracleLoraLinear.forward = fwd


def step(m, wr, data, ac):
    x0, noise, t, ehs, added = data
    for p in m.parameters():
        p.grad = None
    xt = add_noise(x0, noise, t, ddpm_alphas_cumprod())
    if ac:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pred = m(xt, t, ehs, added_cond_kwargs=added).sample
    else:
        pred = m(xt, t, ehs, added_cond_kwargs=added).sample
    loss = F.mse_loss(pred.float(), noise)
    loss.backward()
    g = torch.cat([p.grad.flatten().double() for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)])
    return pred.detach().float(), g


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    modes = [("A reference mode (all fp32)", {"to_out", "ff_proj", "ff_out"}), ("B every output bf16", set()),
             ("C stream writers fp32", {"to_out", "ff_out"}), ("D GEGLU input fp32 only", {"ff_proj"})]
    acc = {name: [0.0, 0.0] for name, _ in modes}
    for s in range(seeds):
        m, wr = build(1 + s)
        g = torch.Generator().manual_seed(100 + s)
        data = (torch.randn(2, 4, 32, 32, generator=g), torch.randn(2, 4, 32, 32, generator=g), torch.tensor([91, 707]),
                torch.randn(2, 77, 64, generator=g), dict(text_embeds=torch.randn(2, 64, generator=g), time_ids=torch.tensor([[256.0, 256.0, 0, 0, 256.0, 256.0]] * 2)))
        OracleLoraLinear.forward = _fwd
        p_ref, g_ref = step(m, wr, data, False)
        for name, cls in modes:
            set_mode(cls)
            p, gg = step(m, wr, data, True)
            e_pred = ((p - p_ref).norm() / p_ref.norm()).item()
            one_m_cos = 1.0 - F.cosine_similarity(gg, g_ref, dim=0).item()
            acc[name][0] += e_pred / seeds; acc[name][1] += one_m_cos / seeds
    OracleLoraLinear.forward = _fwd
    a = acc[modes[0][0]]
    print(f"{'mode':34s} {'pred rel-L2':>12s} {'x A':>6s} {'grad 1-cos':>12s} {'x A':>6s}   (mean of {seeds} seeds)")
    for name, _ in modes:
        e, c = acc[name]
        print(f"{name:34s} {e:12.4e} {e / a[0]:6.2f} {c:12.4e} {c / a[1]:6.2f}")


if __name__ == "__main__":
    main()
