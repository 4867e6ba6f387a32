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
    if len(sys.argv) > 1 and sys.argv[1] == "ckpt":
        ks = ref_lora_ckpt_fixture(GOLD)
        print(len(ks), "lora tensors;", os.path.getsize(os.path.join(GOLD, "ref_lora_unet-7.safetensors")), "bytes")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "minsnr":
        torch.save(minsnr_reference_vectors(), os.path.join(GOLD, "minsnr_reference.pt"))
        print("minsnr_reference.pt", os.path.getsize(os.path.join(GOLD, "minsnr_reference.pt")))
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
