"""Generates tests/golden/* (run in the build container, where /root/reference exists; the fixtures travel, the
reference does not):

  sd15_struct.json       parameter names + shapes parsed from the reference's own structure dump
                         cfgs/unet_struct.txt (the only thing in the reference that pins the UNet).
  lora_reference.pt      inputs/outputs/gradients of the reference's REAL LoRA code (LoraLayer.wrap_model ->
                         LoraPatchContainer.forward/backward, imported unmodified through oracle/ref_shims.py)
                         on seeded inputs, for a bias / no-bias Linear and a whole attention module.
  tiny_unet_oracle.pt    oracle outputs for the TINY config (regression pin of oracle + native model).

    python -m oracle.make_golden
"""
import json
import os
import re
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def parse_unet_struct(path):
    """{param name: shape} from the printed module tree (Conv2d / Linear / GroupNorm / LayerNorm lines)."""
    shapes = {}
    stack = []
    for line in open(path):
        indent = (len(line) - len(line.lstrip())) // 2
        m = re.match(r"\s*\((\w+)\): (\w+)\((.*)$", line)
        if not m:
            continue
        name, kind, rest = m.groups()
        stack = stack[:indent - 1] + [name]
        full = ".".join(stack)
        if kind == "Conv2d":
            cin, cout, kh, kw = map(int, re.match(r"(\d+), (\d+), kernel_size=\((\d+), (\d+)\)", rest).groups())
            shapes[full + ".weight"] = [cout, cin, kh, kw]
            shapes[full + ".bias"] = [cout]
        elif kind == "Linear":
            fin, fout, bias = re.match(r"in_features=(\d+), out_features=(\d+), bias=(\w+)", rest).groups()
            shapes[full + ".weight"] = [int(fout), int(fin)]
            if bias == "True":
                shapes[full + ".bias"] = [int(fout)]
        elif kind == "GroupNorm":
            g, c = map(int, re.match(r"(\d+), (\d+)", rest).groups())
            shapes[full + ".weight"] = [c]; shapes[full + ".bias"] = [c]
        elif kind == "LayerNorm":
            c = int(re.match(r"\((\d+),\)", rest).group(1))
            shapes[full + ".weight"] = [c]; shapes[full + ".bias"] = [c]
        elif kind == "Embedding":
            n, c = map(int, re.match(r"(\d+), (\d+)", rest).groups())
            shapes[full + ".weight"] = [n, c]
    return shapes


def lora_reference_vectors():
    from oracle.ref_shims import load_reference_lora
    layers, plugin = load_reference_lora()
    LoraLayer = layers.LoraLayer
    out = {}
    g = torch.Generator().manual_seed(1234)

    def rnd(*s, scale=1.0):
        return torch.randn(*s, generator=g) * scale

    for tag, (fin, fout, bias, rank, alpha) in {"linear_bias_r4": (48, 40, True, 4, 1.0), "linear_nobias_r8": (64, 96, False, 8, 2.0)}.items():
        parent = nn.Module()
        parent.fc = nn.Linear(fin, fout, bias=bias)
        with torch.no_grad():
            parent.fc.weight.copy_(rnd(fout, fin, scale=fin ** -0.5))
            if bias:
                parent.fc.bias.copy_(rnd(fout, scale=0.1))
        host_w = parent.fc.weight.detach().clone(); host_b = parent.fc.bias.detach().clone() if bias else None
        blocks = LoraLayer.wrap_model(0, parent.fc, parent_block=parent, host_name="fc", rank=rank, alpha=alpha, dropout=0.0)
        blk = blocks[""]
        assert type(parent.fc).__name__ == "LoraPatchContainer"
        with torch.no_grad():
            blk.layer.W_down.copy_(rnd(rank, fin, scale=0.3)); blk.layer.W_up.copy_(rnd(fout, rank, scale=0.3))
        x = rnd(3, 5, fin).requires_grad_(True); dy = rnd(3, 5, fout)
        y = parent.fc(x)
        y.backward(dy)
        out[tag] = dict(host_weight=host_w, host_bias=host_b, W_down=blk.layer.W_down.detach().clone(), W_up=blk.layer.W_up.detach().clone(),
                        alpha_buffer=blk.alpha.clone(), cfg_alpha=alpha, rank=rank, x=x.detach().clone(), dy=dy, y=y.detach().clone(),
                        dx=x.grad.clone(), dW_down=blk.layer.W_down.grad.clone(), dW_up=blk.layer.W_up.grad.clone(),
                        state_keys=sorted(parent.state_dict().keys()))
    # a whole attention module wrapped the way make_hcpdiff does it (wrap_model on the matched `attn` module)
    from oracle.unet_sd15 import CrossAttention
    parent = nn.Module()
    parent.attn2 = CrossAttention(80, 64, 2)
    with torch.no_grad():
        for p in parent.attn2.parameters():
            p.copy_(rnd(*p.shape, scale=p.shape[-1] ** -0.5))
    host_sd = {k: v.clone() for k, v in parent.attn2.state_dict().items()}
    blocks = LoraLayer.wrap_model(0, parent.attn2, parent_block=parent, host_name="attn2", rank=4, alpha=1.0, dropout=0.0)
    with torch.no_grad():
        for b in blocks.values():
            b.layer.W_down.copy_(rnd(*b.layer.W_down.shape, scale=0.2)); b.layer.W_up.copy_(rnd(*b.layer.W_up.shape, scale=0.2))
    x = rnd(2, 16, 80).requires_grad_(True); ctx = rnd(2, 7, 64); dy = rnd(2, 16, 80)
    y = parent.attn2(x, ctx)
    y.backward(dy)
    out["attn2_r4"] = dict(host_state=host_sd, lora={k: dict(W_down=b.layer.W_down.detach().clone(), W_up=b.layer.W_up.detach().clone(),
                                                             dW_down=b.layer.W_down.grad.clone(), dW_up=b.layer.W_up.grad.clone())
                                                     for k, b in blocks.items()},
                           x=x.detach().clone(), ctx=ctx, dy=dy, y=y.detach().clone(), dx=x.grad.clone(),
                           state_keys=sorted(parent.state_dict().keys()))
    # conv hosts (LoCon, cfgs/train/examples/locon.yaml): LoraLayer.Conv2dLayer, lora_layers_patch.py:64-100
    for tag, (cin, cout, stride, rank, alpha, hw) in {"conv3x3_r4": (16, 24, 1, 4, 1.0, 6), "conv3x3_s2_r8": (8, 16, 2, 8, 2.0, 8)}.items():
        parent = nn.Module()
        parent.conv = nn.Conv2d(cin, cout, 3, stride, 1)
        with torch.no_grad():
            parent.conv.weight.copy_(rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5)); parent.conv.bias.copy_(rnd(cout, scale=0.1))
        host_w = parent.conv.weight.detach().clone(); host_b = parent.conv.bias.detach().clone()
        blocks = LoraLayer.wrap_model(0, parent.conv, parent_block=parent, host_name="conv", rank=rank, alpha=alpha, dropout=0.0)
        blk = blocks[""]
        assert type(parent.conv).__name__ == "LoraPatchContainer" and tuple(blk.layer.W_down.shape) == (rank, cin, 3, 3)
        with torch.no_grad():
            blk.layer.W_down.copy_(rnd(rank, cin, 3, 3, scale=0.2)); blk.layer.W_up.copy_(rnd(cout, rank, 1, 1, scale=0.3))
        x = rnd(2, cin, hw, hw).requires_grad_(True)
        y = parent.conv(x)
        dy = rnd(*y.shape)
        y.backward(dy)
        out[tag] = dict(host_weight=host_w, host_bias=host_b, W_down=blk.layer.W_down.detach().clone(), W_up=blk.layer.W_up.detach().clone(),
                        alpha_buffer=blk.alpha.clone(), cfg_alpha=alpha, rank=rank, stride=stride, x=x.detach().clone(), dy=dy,
                        y=y.detach().clone(), dx=x.grad.clone(), dW_down=blk.layer.W_down.grad.clone(), dW_up=blk.layer.W_up.grad.clone(),
                        state_keys=sorted(parent.state_dict().keys()))
    return out


def tiny_unet_vectors():
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import OracleUNet2DConditionModel, TINY_CONFIG, add_noise, ddpm_alphas_cumprod, seeded_init_
    torch.manual_seed(0)
    m = seeded_init_(OracleUNet2DConditionModel(**TINY_CONFIG), 1)
    m.requires_grad_(False)
    wr = wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=4)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for w in wr.values():
            w.lora_block_0.layer.W_up.copy_(torch.randn(w.lora_block_0.layer.W_up.shape, generator=g) * 0.05)
    g2 = torch.Generator().manual_seed(7)
    x0 = torch.randn(2, 4, 8, 8, generator=g2); ehs = torch.randn(2, 77, 64, generator=g2); noise = torch.randn(2, 4, 8, 8, generator=g2)
    t = torch.tensor([10, 500])
    pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs).sample
    loss = F.mse_loss(pred, noise)
    loss.backward()
    grads = torch.cat([p.grad.flatten() for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)])
    return dict(x0=x0, ehs=ehs, noise=noise, t=t, pred=pred.detach(), loss=loss.detach(), lora_grads=grads, n_lora=len(wr))


def grad_fingerprint(named_grads, seed=77):
    """Size-independent summary of a set of gradient tensors: per-tensor L2 norm and projection on a seeded direction."""
    import zlib
    out = {}
    for name, g in named_grads:
        gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        d = torch.randn(g.shape, generator=gen)
        out[name] = (float(g.float().cpu().norm()), float((g.float().cpu() * d).sum() / d.norm()))
    return out


def sd15_full_inputs():
    g2 = torch.Generator().manual_seed(42)
    x0 = torch.randn(1, 4, 64, 64, generator=g2); ehs = torch.randn(1, 77, 768, generator=g2)
    noise = torch.randn(1, 4, 64, 64, generator=g2); t = torch.tensor([437])
    return x0, ehs, noise, t


def sd15_lora_init_(named_lora_params, seed=5):
    """Seeded non-zero LoRA factors by parameter name (W_down ~ N(0, 1/in), W_up ~ 0.05 N(0,1))."""
    import zlib
    with torch.no_grad():
        for name, p in named_lora_params:
            gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            scale = 0.05 if name.endswith("W_up") else p.shape[1] ** -0.5
            p.copy_(torch.randn(p.shape, generator=gen) * scale)


def sd15_full_vectors():
    """Full SD1.5 architecture (859.5 M params), batch 1: oracle prediction, loss and LoRA-gradient fingerprint."""
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod, seeded_init_
    with torch.device("meta"):
        m = OracleUNet2DConditionModel()
    m = seeded_init_(m.to_empty(device="cpu"), 1)
    m.requires_grad_(False)
    wr = wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=8)
    lora_named = [(n, p) for n, p in m.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    x0, ehs, noise, t = sd15_full_inputs()
    pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs).sample
    loss = F.mse_loss(pred, noise)
    loss.backward()
    return dict(pred=pred.detach(), loss=float(loss), n_lora=len(wr), fingerprint=grad_fingerprint([(n, p.grad) for n, p in lora_named]))


def sd15_b4_inputs():
    """The benchmark batch (BASELINE.json configs[1]): B=4, 64x64 latents, 77x768 context, timesteps of SURVEY.md §8(c)."""
    g2 = torch.Generator().manual_seed(4242)
    x0 = torch.randn(4, 4, 64, 64, generator=g2); ehs = torch.randn(4, 77, 768, generator=g2)
    noise = torch.randn(4, 4, 64, 64, generator=g2); t = torch.tensor([10, 250, 500, 999])
    return x0, ehs, noise, t


SD15_BOUNDARIES = ["conv_in", "down_blocks.0", "down_blocks.1", "down_blocks.2", "down_blocks.3", "mid_block",
                   "up_blocks.0", "up_blocks.1", "up_blocks.2", "up_blocks.3", "conv_norm_out"]


def boundary_sample(name, y_nchw, n=8192, seed=31):
    """A seeded sample of a block-boundary activation [B,C,H,W] (logical NCHW coordinates, so the native channels-last tensors are
    sampled at the same elements): (values fp32 [n], L2 norm of the whole tensor)."""
    import zlib
    y = y_nchw.detach().float().cpu()
    gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
    idx = torch.randint(0, y.numel(), (min(n, y.numel()),), generator=gen)
    return y.reshape(-1)[idx].clone(), float(y.norm())


def quantize_grads(named_grads):
    """int8 with one absmax scale per tensor: a 3 MB fixture of the FULL flat LoRA gradient (cosine error of the code ~1e-5)."""
    q, scales = [], []
    for _, g in named_grads:
        g = g.detach().float().cpu().flatten()
        s_ = float(g.abs().max()) / 127.0 or 1.0
        q.append(torch.clamp((g / s_).round(), -127, 127).to(torch.int8)); scales.append(s_)
    return torch.cat(q), torch.tensor(scales)


def dequantize_grads(q, scales, named_params):
    out, off = [], 0
    for (_, p), s_ in zip(named_params, scales.tolist()):
        n = p.numel()
        out.append(q[off:off + n].float() * s_); off += n
    return torch.cat(out)


def sd15_full_b4_vectors():
    """Full SD1.5 architecture at the BENCHMARK batch (B=4): prediction, loss, a sample of every block-boundary activation and the
    full flat LoRA gradient (int8, per-tensor scale) of the fp32 oracle."""
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod, seeded_init_
    with torch.device("meta"):
        m = OracleUNet2DConditionModel()
    m = seeded_init_(m.to_empty(device="cpu"), 1)
    m.requires_grad_(False)
    wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=8)
    lora_named = [(n, p) for n, p in m.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    x0, ehs, noise, t = sd15_b4_inputs()
    named = dict(m.named_modules())
    bounds, hooks = {}, []
    for name in SD15_BOUNDARIES:
        def hook(mod, args, out, name=name):
            y = out[0] if isinstance(out, tuple) else out
            bounds[name] = boundary_sample(name, y)
        hooks.append(named[name].register_forward_hook(hook))
    pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs).sample
    for h in hooks:
        h.remove()
    loss = F.mse_loss(pred, noise)
    loss.backward()
    q, scales = quantize_grads([(n, p.grad) for n, p in lora_named])
    return dict(pred=pred.detach(), loss=float(loss), boundaries=bounds, grad_q=q, grad_scales=scales, grad_names=[n for n, _ in lora_named],
                grad_norm=float(torch.cat([p.grad.flatten() for _, p in lora_named]).norm()))


def sdxl_full_inputs():
    """SDXL-base shapes at batch 1 / 512 px (64x64 latents keep the CPU oracle to minutes; every layer shape except the
    token count equals BASELINE.json configs[3])."""
    g2 = torch.Generator().manual_seed(43)
    x0 = torch.randn(1, 4, 64, 64, generator=g2); ehs = torch.randn(1, 77, 2048, generator=g2)
    noise = torch.randn(1, 4, 64, 64, generator=g2); t = torch.tensor([611])
    added = dict(text_embeds=torch.randn(1, 1280, generator=g2), time_ids=torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]]))
    return x0, ehs, noise, t, added


def sdxl_full_vectors():
    """Full SDXL-base architecture (2.567 B params, seeded init), LoRA rank 16 on attn/ff blocks."""
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import OracleUNet2DConditionModel, SDXL_CONFIG, add_noise, ddpm_alphas_cumprod, seeded_init_
    with torch.device("meta"):
        m = OracleUNet2DConditionModel(**SDXL_CONFIG)
    m = seeded_init_(m.to_empty(device="cpu"), 1)
    m.requires_grad_(False)
    wr = wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)
    lora_named = [(n, p) for n, p in m.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    x0, ehs, noise, t, added = sdxl_full_inputs()
    pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs, added_cond_kwargs=added).sample
    loss = F.mse_loss(pred, noise)
    loss.backward()
    return dict(pred=pred.detach(), loss=float(loss), n_lora=len(wr), n_lora_params=sum(p.numel() for _, p in lora_named),
                fingerprint=grad_fingerprint([(n, p.grad) for n, p in lora_named]))


def tensor_sketch(name, g, n=4096, seed=91):
    """Per-tensor record of a gradient that is too large to ship whole (full fine-tune: 859.5 M elements): L2 norm, and a seeded
    sample of min(numel, n) elements at LOGICAL (row-major over the parameter's shape) positions, int16 with one absmax scale."""
    import zlib
    g = g.detach().float().cpu().contiguous().flatten()
    gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
    idx = torch.arange(g.numel()) if g.numel() <= n else torch.randint(0, g.numel(), (n,), generator=gen)
    v = g[idx]
    s_ = float(v.abs().max()) / 32767.0 or 1.0
    return dict(norm=float(g.norm()), scale=s_, q=torch.clamp((v / s_).round(), -32767, 32767).to(torch.int16))


def sketch_values(name, g, n=4096, seed=91):
    """The same sample of a live tensor, fp32 (test side)."""
    import zlib
    g = g.detach().float().cpu().contiguous().flatten()
    gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
    idx = torch.arange(g.numel()) if g.numel() <= n else torch.randint(0, g.numel(), (n,), generator=gen)
    return g[idx]


def _full_oracle(cfg=None, seed=1):
    from oracle import unet_sd15 as U
    U.ATTN_RECOMPUTE = True                         # identical arithmetic; the N x N score tensors are not kept for backward
    with torch.device("meta"):
        m = U.OracleUNet2DConditionModel(**(cfg or {}))
    return U.seeded_init_(m.to_empty(device="cpu"), seed)


def dreambooth_inputs():
    """BASELINE.json configs[2] as cfgs/train/examples/DreamBooth.yaml lays the step out: TWO datasets per optimisation step —
    the instance batch (bs 2, the configuration's batch size) and the class / regularisation batch (DreamBooth.yaml:47-48:
    batch_size 1, loss_weight 1.0) — 512 px (64x64 latents), 77x768 context."""
    g2 = torch.Generator().manual_seed(4343)
    out = []
    for B, ts in ((2, [37, 803]), (1, [512])):
        out.append(dict(x0=torch.randn(B, 4, 64, 64, generator=g2), ehs=torch.randn(B, 77, 768, generator=g2),
                        noise=torch.randn(B, 4, 64, 64, generator=g2), t=torch.tensor(ts), loss_weight=1.0))
    return out


def dreambooth_b2_vectors():
    """SD1.5 full fine-tune (every one of the 686 parameter tensors / 859.5 M elements trainable, DreamBooth.yaml:6-10), one
    optimisation step's gradient = instance batch + class batch (train_ac.py:467-483): fp32 oracle predictions and losses per
    dataset, and a sketch (norm + 4096-element seeded sample) of EVERY parameter's accumulated gradient."""
    import torch.nn.functional as F
    from oracle.unet_sd15 import add_noise, ddpm_alphas_cumprod
    m = _full_oracle()
    preds, losses = [], []
    for d in dreambooth_inputs():
        pred = m(add_noise(d["x0"], d["noise"], d["t"], ddpm_alphas_cumprod()), d["t"], d["ehs"]).sample
        loss = F.mse_loss(pred, d["noise"]) * d["loss_weight"]
        loss.backward()
        preds.append(pred.detach()); losses.append(float(loss))
    named = list(m.named_parameters())
    return dict(preds=preds, losses=losses, names=[n for n, _ in named], sketch={n: tensor_sketch(n, p.grad) for n, p in named},
                grad_norm=float(torch.sqrt(sum(p.grad.double().pow(2).sum() for _, p in named))))


def sdxl_b2_inputs():
    """BASELINE.json configs[3] at its REAL shape: SDXL-base, bs 2, 1024 px = 128x128 latents, 77x2048 context, pooled 1280,
    crop_info = [1024, 1024, 0, 0, 1024, 1024]."""
    g2 = torch.Generator().manual_seed(4444)
    x0 = torch.randn(2, 4, 128, 128, generator=g2); ehs = torch.randn(2, 77, 2048, generator=g2)
    noise = torch.randn(2, 4, 128, 128, generator=g2); t = torch.tensor([91, 707])
    added = dict(text_embeds=torch.randn(2, 1280, generator=g2), time_ids=torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]] * 2))
    return x0, ehs, noise, t, added


def sdxl_b2_draw_inputs(draw):
    """Inputs of the configs[3] shape for another DRAW (round 6: the parity ratios of one fixture are one draw of quantities that scatter,
    DESIGN section 4): draw 0 = sdxl_b2_inputs(); draw d > 0 = latents, prompt / pooled states, noise and timesteps from seed 1000 + d.
    Used by tools/diag/sdxl_grad_draws.py (which writes tests/golden/sdxl_b2_draw<d>_oracle.pt on the GPU box's host cores) and by the test."""
    x0, ehs, noise, t, added = sdxl_b2_inputs()
    if draw:
        g3 = torch.Generator().manual_seed(1000 + draw)
        x0 = torch.randn(x0.shape, generator=g3); ehs = torch.randn(ehs.shape, generator=g3); noise = torch.randn(noise.shape, generator=g3)
        t = torch.randint(0, 1000, t.shape, generator=g3)
        added = dict(text_embeds=torch.randn(added["text_embeds"].shape, generator=g3), time_ids=added["time_ids"])
    return x0, ehs, noise, t, added


def sdxl_b2_vectors():
    """Full SDXL-base (2.567 B parameters), LoRA rank 16 on attn / ff (700 layers, 41,861,120 LoRA parameters): prediction, loss and
    the ENTIRE flat LoRA gradient (int8, one absmax scale per tensor) of the fp32 oracle with the reference-form merged-weight LoRA."""
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import SDXL_CONFIG, add_noise, ddpm_alphas_cumprod
    m = _full_oracle(SDXL_CONFIG)
    m.requires_grad_(False)
    wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)
    lora_named = [(n, p) for n, p in m.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    x0, ehs, noise, t, added = sdxl_b2_inputs()
    pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs, added_cond_kwargs=added).sample
    loss = F.mse_loss(pred, noise)
    loss.backward()
    q, scales = quantize_grads([(n, p.grad) for n, p in lora_named])
    return dict(pred=pred.detach().half(), loss=float(loss), grad_q=q, grad_scales=scales, grad_names=[n for n, _ in lora_named],
                grad_norm=float(torch.cat([p.grad.flatten() for _, p in lora_named]).norm()))


def lora_tensor_class(name):
    """(resolution block, layer kind) of a LoRA parameter name: the granularity at which tests/test_full_configs.py compares the
    native error with the reference-under-autocast error."""
    import re
    blk = re.match(r"(down_blocks\.\d+|mid_block|up_blocks\.\d+)", name)
    kind = re.search(r"(attn1\.to_q|attn1\.to_k|attn1\.to_v|attn1\.to_out\.0|attn2\.to_q|attn2\.to_k|attn2\.to_v|attn2\.to_out\.0|ff\.net\.0\.proj|ff\.net\.2)", name)
    side = "W_down" if name.endswith("W_down") else "W_up"
    return f"{blk.group(1) if blk else '?'}|{kind.group(1) if kind else '?'}|{side}"


def sdxl_b2_autocast_calibration():
    """how far is the REFERENCE's own execution mode from the fp32 oracle?  The reference trains under
    torch.autocast(bfloat16) with fp32 parameters (train_ac.py:449 `with torch.autocast(...)`): matmuls / convolutions and the residual
    stream in bf16.  The same oracle graph, seeds and inputs as sdxl_b2_vectors() are run under torch.autocast("cpu", bfloat16) and
    compared with the committed fp32 fixture: prediction rel-L2, flat / per-tensor / per-class LoRA-gradient cosine.  The native
    test then asserts its own distance from fp32 is at most 1.25 x this one (instead of a hand-set 0.998 / 0.985)."""
    import torch.nn.functional as F
    from oracle.lora_ref import wrap_lora
    from oracle.unet_sd15 import SDXL_CONFIG, add_noise, ddpm_alphas_cumprod
    g = torch.load(os.path.join(GOLD, "sdxl_full_b2_oracle.pt"))
    m = _full_oracle(SDXL_CONFIG)
    m.requires_grad_(False)
    wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)
    lora_named = [(n, p) for n, p in m.named_parameters() if "lora_block_" in n]
    assert [n for n, _ in lora_named] == g["grad_names"]
    sd15_lora_init_(lora_named)
    x0, ehs, noise, t, added = sdxl_b2_inputs()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        pred = m(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs, added_cond_kwargs=added).sample
        loss = F.mse_loss(pred.float(), noise)
    loss.backward()
    ref = dequantize_grads(g["grad_q"], g["grad_scales"], lora_named).double()
    flat = torch.cat([p.grad.detach().float().flatten() for _, p in lora_named]).double()
    per_tensor, cls_acc, off = [], {}, 0
    for n, p in lora_named:
        a, b = flat[off:off + p.numel()], ref[off:off + p.numel()]; off += p.numel()
        per_tensor.append(float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300)))
        acc = cls_acc.setdefault(lora_tensor_class(n), [0.0, 0.0, 0.0])
        acc[0] += float(a @ b); acc[1] += float(a @ a); acc[2] += float(b @ b)
    ref_pred = g["pred"].float()
    return dict(names=g["grad_names"], per_tensor_cos=torch.tensor(per_tensor), flat_cos=float(flat @ ref / (flat.norm() * ref.norm())),
                class_cos={k: v[0] / max((v[1] * v[2]) ** 0.5, 1e-300) for k, v in cls_acc.items()},
                pred_rel=float((pred.detach().float() - ref_pred).norm() / ref_pred.norm()), loss=float(loss), loss_fp32=g["loss"],
                grad_norm=float(flat.norm()), note="oracle under torch.autocast('cpu', bfloat16) vs the fp32 oracle fixture (int8 gradient code, cosine error ~1e-5)")


def controlnet_b4_inputs():
    """BASELINE.json configs[4]: frozen SD1.5 + ControlNet branch, bs 4, 512 px; control image [4,3,512,512] ~ U[0,1]."""
    g2 = torch.Generator().manual_seed(4545)
    x0 = torch.randn(4, 4, 64, 64, generator=g2); ehs = torch.randn(4, 77, 768, generator=g2)
    noise = torch.randn(4, 4, 64, 64, generator=g2); t = torch.tensor([10, 250, 500, 999])
    cond = torch.rand(4, 3, 512, 512, generator=g2)
    return x0, ehs, noise, t, cond


def controlnet_init_(ocn, seed=8):
    """Seeded non-zero values for the branch's zero convs and cond_head (a branch 'after some training': with the reference's zero
    init every gradient upstream of the zero convs is exactly zero), by parameter name."""
    import zlib
    with torch.no_grad():
        for n_, p_ in ocn.named_parameters():
            if n_.startswith(("cond_head", "controlnet_")):
                gen = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(n_.encode())) % (2 ** 31))
                p_.copy_(torch.randn(p_.shape, generator=gen) * (0.3 / max(1.0, p_[0].numel() ** 0.5) if p_.dim() > 1 else 0.05))


def controlnet_b4_vectors():
    """Full SD1.5 host (frozen) + full ControlNet branch (361 M trainable parameters: deep copy of the encoder, cond_head, 13 zero
    convs — reference controlnet.py:11-62), one step at bs 4: prediction, loss, a sample of the 13 residuals' norms, and a sketch of
    EVERY branch parameter's gradient (fp32 oracle restatement, itself pinned to the reference's plugin code at small size)."""
    import torch.nn.functional as F
    from oracle.unet_sd15 import OracleControlNet, add_noise, ddpm_alphas_cumprod
    m = _full_oracle()
    m.requires_grad_(False)
    torch.manual_seed(3)
    ocn = OracleControlNet(m)
    for p_ in ocn.parameters():
        p_.requires_grad_(True)
    controlnet_init_(ocn)
    x0, ehs, noise, t, cond = controlnet_b4_inputs()
    xt = add_noise(x0, noise, t, ddpm_alphas_cumprod())
    res = ocn(xt, t, ehs, cond)
    pred = m(xt, t, ehs, control_residuals=res).sample
    loss = F.mse_loss(pred, noise)
    loss.backward()
    named = list(ocn.named_parameters())
    return dict(pred=pred.detach(), loss=float(loss), residual_norms=[float(r.detach().norm()) for r in res], names=[n for n, _ in named],
                sketch={n: tensor_sketch(n, p.grad) for n, p in named},
                grad_norm=float(torch.sqrt(sum(p.grad.double().pow(2).sum() for _, p in named))))


def reference_trainer_trajectory():
    """Ten optimisation steps of the reference's OWN inner loop — hcpdiff/train_ac.py Trainer.train_one_step / forward / make_noise /
    get_loss (train_ac.py:437-515) on a TrainerSingleCard with a real accelerate.Accelerator, its TEUnetWrapper (models/wrapper.py),
    its make_hcpdiff + LoraLayer (the reference's own LoRA, type 'lora', dropout 0) and torch.optim.AdamW — over the fp32 ORACLE UNet
    and text encoder (the un-vendored diffusers / transformers arithmetic), BASELINE.json configs[0] in miniature (rank 4 on attn + ff,
    CPU, fp32).  No native code runs: this is the reference side of the parity test that NativeTrainer must reproduce on the GPU.
    Recorded: the data, the noise / timesteps the reference drew (its torch CPU RNG stream cannot be reproduced on a device), the ten
    losses, the initial and final LoRA factors."""
    import types
    from oracle.ref_shims import load_reference_trainer
    train_ac, single = load_reference_trainer()
    import hcpdiff.utils.cfg_net_tools as tools
    from hcpdiff.models import CFGContext, TEUnetWrapper
    from oracle.clip_ref import OracleCLIPTextModel
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod, seeded_init_
    UCFG = dict(MICRO_CONFIG, cross_attention_dim=64)
    TCFG = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    STEPS, B = 10, 2
    u = seeded_init_(OracleUNet2DConditionModel(**UCFG), 1); u.requires_grad_(False); u.eval()
    te = seeded_init_(OracleCLIPTextModel(**TCFG), 2); te.requires_grad_(False); te.eval()
    g = torch.Generator().manual_seed(9)
    data = [dict(img=torch.randn(B, 4, 8, 8, generator=g), prompt=torch.randint(0, 100, (B, 77), generator=g)) for _ in range(STEPS)]
    ns = types.SimpleNamespace
    t = single.TrainerSingleCard.__new__(single.TrainerSingleCard)
    t.cfgs = ns(seed=114514, mixed_precision="no", train=ns(gradient_accumulation_steps=1, max_grad_norm=1.0, set_grads_to_none=False, loss=ns(type="eps")))
    t.init_context(None)
    t.weight_dtype = torch.float32
    pats = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
    groups, lora_unet = tools.make_hcpdiff(u, None, [_Item(layers=pats, rank=4, dropout=0.0, lr=1e-3)])
    gi = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for path in sorted(lora_unet.plugin_dict):
            blk = lora_unet.plugin_dict[path]
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=gi) * 0.05)
            blk.layer.W_down.copy_(torch.randn(blk.layer.W_down.shape, generator=gi) * blk.layer.W_down.shape[1] ** -0.5)
    init = {k: (b.layer.W_down.detach().clone(), b.layer.W_up.detach().clone()) for k, b in lora_unet.plugin_dict.items()}
    drawn = []

    class Sched:                                       # the seam-4 object: DDPMScheduler.add_noise [ext], recording what make_noise drew
        config = ns(num_train_timesteps=1000)

        def add_noise(self, latents, noise, timesteps):
            drawn.append((noise.clone(), timesteps.clone()))
            return add_noise(latents, noise, timesteps, ddpm_alphas_cumprod())

    class TE(torch.nn.Module):                         # transformers' CLIPTextModel call surface (wrapper.py:20): output[0] = last_hidden_state
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, input_ids, position_ids=None, attention_mask=None, output_hidden_states=False):
            return (self.m.encode(input_ids, position_ids, attention_mask=attention_mask),)
    t.TE_unet = TEUnetWrapper(u, TE(te))
    t.noise_scheduler = Sched()
    t.cfg_context = CFGContext()
    t.criterion = torch.nn.MSELoss(reduction="none")
    t.embedding_hook = ns(emb_train=[])
    t.train_loader_group = ns(get_dataset=lambda idx: ns(latents=True), get_loss_weights=lambda idx: 1.0)
    t.optimizer = torch.optim.AdamW(groups, weight_decay=1e-3)
    t.lr_scheduler = None
    torch.manual_seed(1234)
    losses = [t.train_one_step([dict(d)]) for d in data]
    final = {k: (b.layer.W_down.detach().clone(), b.layer.W_up.detach().clone()) for k, b in lora_unet.plugin_dict.items()}
    return dict(unet_cfg=UCFG, te_cfg=TCFG, data=data, drawn=drawn, losses=losses, lora_init=init, lora_final=final, lr=1e-3, weight_decay=1e-3,
                source="hcpdiff/train_ac.py Trainer.train_one_step (unmodified) over oracle UNet / CLIP, reference LoraLayer, torch AdamW")


class _Item(dict):
    """The reference reads cfg items both as mappings and as attribute bags (OmegaConf DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def ref_lora_ckpt_fixture(out_dir):
    """A LoRA checkpoint WRITTEN BY THE REFERENCE: its LoraLayer (lora_layers_patch.py) wrapped around the MICRO oracle UNet
    by the make_hcpdiff loop (cfg_net_tools.py:108-123), saved by CkptManagerSafe.save_model_with_lora (ckpt_pkl.py:56-72,
    ckpt_safetensor.py:20-27) -> tests/golden/ref_lora_unet-7.safetensors, and the prediction of that model (the reference's
    LoraPatchContainer forward) on seeded inputs -> tests/golden/ref_lora_ckpt_expect.pt."""
    from oracle.ref_shims import load_reference_ckpt
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_
    CkptManagerSafe, tools = load_reference_ckpt()
    m = seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1)
    m.requires_grad_(False)
    cfg = [_Item(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=4, alpha=2.0), _Item(layers=[r"re:.*\.resnets\.0\.conv1$"], rank=8, alpha=2.0)]
    torch.manual_seed(123)                      # W_down: the reference's kaiming_uniform_ draws from the global generator
    _, group = tools.make_hcpdiff(m, None, cfg)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for blk in group.plugin_dict.values():
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
    mgr = CkptManagerSafe()
    mgr.set_save_dir(out_dir)
    mgr.save_model_with_lora(m, group, name="ref_lora_unet", step=7)
    x = torch.randn(2, 4, 8, 8, generator=g); ehs = torch.randn(2, 77, MICRO_CONFIG["cross_attention_dim"], generator=g)
    t = torch.tensor([30, 700])
    with torch.no_grad():
        pred = m(x, t, ehs).sample
    keys = sorted(group.state_dict().keys())
    torch.save(dict(x=x, ehs=ehs, t=t, pred=pred, keys=keys, cfg=[dict(c) for c in cfg], host_seed=1),
               os.path.join(out_dir, "ref_lora_ckpt_expect.pt"))
    return keys


def vae_full_vectors():
    """Oracle latents of the full-size SD VAE encoder (seeded weights) for one seeded 256 px, 512 px and 1024 px (SDXL-size) image."""
    from oracle.unet_sd15 import seeded_init_
    from oracle.vae_ref import SD_VAE_CONFIG, OracleVAEEncoder
    m = seeded_init_(OracleVAEEncoder(**SD_VAE_CONFIG), 7)
    out = {}
    for side in (256, 512, 1024):
        gen = torch.Generator().manual_seed(100 + side)
        img = torch.rand(1, 3, side, side, generator=gen) * 2 - 1
        noise = torch.randn(1, 4, side // 8, side // 8, generator=gen)
        with torch.no_grad():
            out[side] = dict(seed=7, input_seed=100 + side, latents=m.encode(img, noise))
    return out


def vae_decode_vectors():
    """Oracle images of the full-size SD VAE DECODER (seeded weights) for seeded 32x32 (256 px) and 64x64 (512 px) latents: the whole
    256 px image in fp16 and 16384 seeded pixel samples + the norm of the 512 px image."""
    from oracle.unet_sd15 import seeded_init_
    from oracle.vae_ref import SD_VAE_CONFIG, OracleAutoencoderKL
    m = seeded_init_(OracleAutoencoderKL(**SD_VAE_CONFIG), 9)
    out = {}
    for side in (256, 512):
        gen = torch.Generator().manual_seed(300 + side)
        z = torch.randn(1, 4, side // 8, side // 8, generator=gen)
        with torch.no_grad():
            img = m.decode(z)
        vals, norm = boundary_sample(f"vae_dec_{side}", img, n=16384)
        out[side] = dict(seed=9, input_seed=300 + side, samples=vals, norm=norm, image=img.half() if side == 256 else None)
    return out


CNET_CONFIG = dict(block_out_channels=(16, 32, 32, 32), layers_per_block=2, num_attention_heads=1, cross_attention_dim=16, norm_num_groups=4)


def controlnet_reference_vectors():
    """The reference's OWN ControlNetPlugin (hcpdiff/models/controlnet.py: construction :11-62, hooks :64-82, forward :88-183, hook
    registration of MultiPluginBlock plugin.py:175-201), executed unmodified on top of the oracle UNet — a small UNet with SD1.5's
    block layout (4 down blocks x 2 layers: the plugin's residual indices are hard-coded for it).  The oracle's blocks are given the
    keyword names diffusers uses at the plugin's call sites (hidden_states= / temb= / encoder_hidden_states= ...) for the duration of
    the run; nothing else is adapted.  Output: the UNet prediction with the branch attached, the 13 residuals, and the plugin's
    state_dict (so that OracleControlNet can be loaded with identical weights)."""
    import importlib
    from oracle.ref_shims import load_reference_ckpt
    from oracle import unet_sd15 as U
    _, tools = load_reference_ckpt()
    diffusers = sys.modules["diffusers"]             # the shim package ref_shims registered (the real one is not installed)
    if not hasattr(diffusers, "UNet2DConditionModel"):
        diffusers.UNet2DConditionModel = type("UNet2DConditionModel", (), {})
    cn = importlib.import_module("hcpdiff.models.controlnet")

    def block_fwd(orig):
        def fwd(self, *a, hidden_states=None, temb=None, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None):
            a = list(a)
            h = hidden_states if hidden_states is not None else a.pop(0)
            t = temb if temb is not None else a.pop(0)
            c = encoder_hidden_states if encoder_hidden_states is not None else (a.pop(0) if a else None)
            return orig(self, h, t, c)
        return fwd

    saved = {cls: cls.forward for cls in (U.DownBlock, U.MidBlock, U.TimestepEmbedding)}
    U.DownBlock.forward, U.MidBlock.forward = block_fwd(saved[U.DownBlock]), block_fwd(saved[U.MidBlock])
    U.TimestepEmbedding.forward = lambda self, x, cond=None, _o=saved[U.TimestepEmbedding]: _o(self, x)
    try:
        host = U.seeded_init_(U.OracleUNet2DConditionModel(**CNET_CONFIG), 1)
        host.class_embedding, host.dtype = None, torch.float32
        for blk in host.down_blocks:
            blk.has_cross_attention = blk.has_attn
        named = dict(host.named_modules())
        metas = lambda pats: [{**m, "layer": named[m["layer"]]} for m in tools.get_match_layers(pats, named, return_metas=True)]
        plug = cn.ControlNetPlugin("controlnet1", metas(["pre_hook:", "pre_hook:conv_in"]),
                                   metas([f"down_blocks.{i}" for i in range(4)] + ["mid_block", "pre_hook:up_blocks.3.resnets.2"]),
                                   host_model=host, cond_block_channels=(3, 4, 8, 8, 16, 16), layers_per_block=2, block_out_channels=CNET_CONFIG["block_out_channels"])
        g = torch.Generator().manual_seed(21)
        with torch.no_grad():                       # non-zero "zero convs" and a branch that differs from the host, as after training
            for n, p in plug.named_parameters():
                if n.startswith(("controlnet_", "cond_head")):
                    p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.05))
                else:
                    p.add_(torch.randn(p.shape, generator=g) * 0.02)
        x = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([40, 900]); ehs = torch.randn(2, 10, 16, generator=g)
        cond = torch.rand(2, 3, 64, 64, generator=g)
        for feeder in host.input_feeder:
            feeder(dict(cond=cond))
        with torch.no_grad():
            pred = host(x, t, ehs).sample
            residuals = [r.clone() for r in plug.feat_to]
        sd = {k: v.clone() for k, v in plug.state_dict().items()}
        plug.remove()
        with torch.no_grad():
            pred_plain = host(x, t, ehs).sample
        return dict(x=x, t=t, ehs=ehs, cond=cond, pred=pred, pred_without_branch=pred_plain, residuals=residuals, plugin_state=sd, host_seed=1,
                    config=CNET_CONFIG, cond_block_channels=(3, 4, 8, 8, 16, 16))
    finally:
        for cls, f in saved.items():
            cls.forward = f


def minsnr_reference_vectors():
    """Outputs of the REFERENCE's MinSNRLoss / SoftMinSNRLoss / KDiffMinSNRLoss / EDMLoss (min_snr_loss.py) wrapped in
    Trainer.get_loss's reduction (train_ac.py:506-515) on seeded inputs, SD beta schedule."""
    from oracle.ref_shims import load_reference_loss
    from oracle.loss_ref import KINDS, REFERENCE_CLASS
    from oracle.unet_sd15 import ddpm_alphas_cumprod
    ref = load_reference_loss()
    g = torch.Generator().manual_seed(31)
    pred = torch.randn(6, 4, 8, 8, generator=g); target = torch.randn(6, 4, 8, 8, generator=g)
    mask = (torch.rand(6, 1, 8, 8, generator=g) > 0.25).float()
    t = torch.tensor([0, 17, 250, 500, 871, 999], dtype=torch.int64)
    sched = type("Sched", (), {"alphas_cumprod": ddpm_alphas_cumprod()})()
    out = dict(pred=pred, target=target, mask=mask, timesteps=t, cases={})
    for kind in KINDS:
        for gamma in (1.0, 5.0):
            crit = getattr(ref, REFERENCE_CLASS[kind])(gamma=gamma, noise_scheduler=sched, device="cpu")
            assert crit.need_timesteps
            pr = pred.clone().requires_grad_(True)
            per = crit(pr.float(), target.float(), t)                        # train_ac.py:510
            loss = (per * mask).mean()
            loss.backward()
            out["cases"][(kind, gamma)] = dict(loss=float(loss.detach()), grad=pr.grad.clone(), weight=(per.detach() / ((pred - target) ** 2))[:, 0, 0, 0].clone())
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "controlnet":
        torch.save(controlnet_reference_vectors(), os.path.join(GOLD, "controlnet_reference.pt"))
        print("controlnet_reference.pt", os.path.getsize(os.path.join(GOLD, "controlnet_reference.pt")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "te_struct":
        ref = os.environ.get("HCP_REFERENCE_ROOT", "/root/reference")
        shapes = parse_unet_struct(os.path.join(ref, "cfgs", "te_struct.txt"))
        json.dump({"source": "reference cfgs/te_struct.txt", "n_params": sum(int(torch.tensor(s).prod()) for s in shapes.values()),
                   "shapes": shapes}, open(os.path.join(GOLD, "te_struct.json"), "w"), indent=0)
        print("te_struct.json:", len(shapes), "tensors")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vae":
        torch.save(vae_full_vectors(), os.path.join(GOLD, "vae_full_oracle.pt"))
        print("vae_full_oracle.pt", os.path.getsize(os.path.join(GOLD, "vae_full_oracle.pt")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vae_dec":
        torch.save(vae_decode_vectors(), os.path.join(GOLD, "vae_decode_oracle.pt"))
        print("vae_decode_oracle.pt", os.path.getsize(os.path.join(GOLD, "vae_decode_oracle.pt")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ckpt":
        ks = ref_lora_ckpt_fixture(GOLD)
        print(len(ks), "lora tensors;", os.path.getsize(os.path.join(GOLD, "ref_lora_unet-7.safetensors")), "bytes")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "minsnr":
        torch.save(minsnr_reference_vectors(), os.path.join(GOLD, "minsnr_reference.pt"))
        print("minsnr_reference.pt", os.path.getsize(os.path.join(GOLD, "minsnr_reference.pt")))
        sys.exit(0)
    for key, fn, fname in (("dreambooth", dreambooth_b2_vectors, "sd15_dreambooth_b2_oracle.pt"), ("sdxl_b2", sdxl_b2_vectors, "sdxl_full_b2_oracle.pt"),
                           ("sdxl_b2_autocast", sdxl_b2_autocast_calibration, "sdxl_b2_autocast_calibration.pt"),
                           ("controlnet_b4", controlnet_b4_vectors, "sd15_controlnet_b4_oracle.pt"), ("trainer", reference_trainer_trajectory, "ref_trainer_trajectory.pt")):
        if len(sys.argv) > 1 and sys.argv[1] == key:
            import time
            t0 = time.time()
            torch.save(fn(), os.path.join(GOLD, fname))
            print(fname, os.path.getsize(os.path.join(GOLD, fname)), f"{time.time() - t0:.0f} s")
            sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sdxl":
        torch.save(sdxl_full_vectors(), os.path.join(GOLD, "sdxl_full_oracle.pt"))
        print("sdxl_full_oracle.pt", os.path.getsize(os.path.join(GOLD, "sdxl_full_oracle.pt")))
        sys.exit(0)
    os.makedirs(GOLD, exist_ok=True)
    ref = os.environ.get("HCP_REFERENCE_ROOT", "/root/reference")
    shapes = parse_unet_struct(os.path.join(ref, "cfgs", "unet_struct.txt"))
    json.dump({"source": "reference cfgs/unet_struct.txt", "n_params": sum(int(torch.tensor(s).prod()) for s in shapes.values()),
               "shapes": shapes}, open(os.path.join(GOLD, "sd15_struct.json"), "w"), indent=0)
    print("sd15_struct.json:", len(shapes), "tensors")
    torch.save(lora_reference_vectors(), os.path.join(GOLD, "lora_reference.pt"))
    if len(sys.argv) > 1 and sys.argv[1] == "lora":
        sys.exit(0)
    torch.save(tiny_unet_vectors(), os.path.join(GOLD, "tiny_unet_oracle.pt"))
    torch.save(sd15_full_vectors(), os.path.join(GOLD, "sd15_full_oracle.pt"))
    torch.save(sd15_full_b4_vectors(), os.path.join(GOLD, "sd15_full_b4_oracle.pt"))
    for f in os.listdir(GOLD):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
