"""ORACLE (test infrastructure — never imported by the product path; see oracle/README.md).

Plain-PyTorch fp32 CPU restatement of the arithmetic behind the reference's hot-path call
``self.unet(noisy_latents, timesteps, encoder_hidden_states).sample`` (reference hcpdiff/models/wrapper.py:29).

The arithmetic lives in the un-vendored dependency ``diffusers<=0.26.1`` (reference requirements.txt:4), which is
absent from this container, so this file restates ``UNet2DConditionModel`` from
  * the complete SD1.5 module tree the reference ships:      cfgs/unet_struct.txt:1-931 (names, channels, eps, bias flags),
  * the reference's own restatement of the encoder forward:   hcpdiff/models/controlnet.py:88-183 (block call
    conventions, skip-tuple layout) and :71-82 (decoder concat order [hidden, skip]),
  * the published diffusers 0.26.1 semantics marked [ext] below (unverifiable offline).
PARITY STATUS: "parity unpinned" for this file — the reference has no tests / golden vectors for the UNet
 and diffusers cannot be imported here.  The LoRA half IS pinned (oracle/lora_ref.py).

Module / parameter names equal diffusers' so state_dicts are interchangeable with the native model.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    num_attention_heads=8, cross_attention_dim=768, norm_num_groups=32, num_train_timesteps=1000,
    transformer_layers_per_block=1, use_linear_projection=False, addition_embed_type=None, addition_time_embed_dim=None,
    projection_class_embeddings_input_dim=None)

# SDXL-base UNet (BASELINE.json configs[3]).  NOT present anywhere in the reference: restated from the
# public SDXL config.json [ext]: 3 levels, transformer depth 1/2/10, head_dim 64 (diffusers' mis-named
# ``attention_head_dim`` = heads per level 5/10/20), cross dim 2048, linear proj_in/out, "text_time" additional embedding
# (6 micro-conditioning scalars x 256 sinusoid + 1280 pooled text = 2816 -> 1280).  The reference only fixes the CALL:
# ``unet(..., added_cond_kwargs={"text_embeds", "time_ids"})`` (hcpdiff/models/wrapper.py:66-73) and the block-index map
# (hcpdiff/tools/lora_convert.py:116-186: attentions at down_blocks.1-2, mid, up_blocks.0-1).
SDXL_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    num_attention_heads=(5, 10, 20), cross_attention_dim=2048, norm_num_groups=32, num_train_timesteps=1000,
    transformer_layers_per_block=(1, 2, 10), use_linear_projection=True, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)

# Miniature with the SDXL structure (head_dim 64; pooled text 64 + 6 x 32 = 256 additional-embedding input).
TINY_SDXL_CONFIG = dict(SDXL_CONFIG, block_out_channels=(64, 128, 256), layers_per_block=1, num_attention_heads=(1, 2, 4),
                        cross_attention_dim=64, norm_num_groups=8, transformer_layers_per_block=(1, 1, 2),
                        addition_time_embed_dim=32, projection_class_embeddings_input_dim=256)


# Two-level miniature (head dims 40 / 80) for the CPU-interpreted tests of the secondary features: same block types, ~4x less work.
MICRO_CONFIG = dict(SD15_CONFIG, block_out_channels=(40, 80), layers_per_block=1, num_attention_heads=1, cross_attention_dim=32,
                    norm_num_groups=8, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def per_block(v, i):
    """diffusers accepts an int or a per-down-block tuple for heads / transformer depth."""
    return v[i] if isinstance(v, (tuple, list)) else v

# A structurally identical miniature (same block types; channel counts chosen so head_dim is 40 / 80 as in SD1.5).
TINY_CONFIG = dict(SD15_CONFIG, block_out_channels=(80, 160, 160, 160), layers_per_block=1, num_attention_heads=2,
                   cross_attention_dim=64, norm_num_groups=8)   # head dims 40/80 = the SD1.5 ones


def timestep_embedding(timesteps, dim, max_period=10000.0):
    """Timesteps(num_channels=320, flip_sin_to_cos=True, downscale_freq_shift=0)  [ext]; unet_struct.txt:3."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)  # flip_sin_to_cos=True -> [cos | sin]


class Timesteps(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.num_channels = dim

    def forward(self, t):
        return timestep_embedding(t, self.num_channels)


class TimestepEmbedding(nn.Module):  # unet_struct.txt:4-8
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ResnetBlock2D(nn.Module):  # unet_struct.txt:92-100, :199-208 (conv_shortcut when Cin != Cout)
    def __init__(self, cin, cout, temb_dim, groups, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]   # [ext] ResnetBlock2D.forward
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if hasattr(self, "conv_shortcut"):
            x = self.conv_shortcut(x)
        return x + h                                                            # output_scale_factor = 1


# Golden generators for the full-size configurations (oracle/make_golden.py) set this: the [B, heads, N, N] score / probability
# tensors of every attention layer are then recomputed in backward instead of kept (same arithmetic, same order: the result is
# bit-identical; SDXL at 128x128 latents would otherwise keep ~70 GB of them).
ATTN_RECOMPUTE = False


def _attention_core(q, k, v, key_bias, scale):
    s = q @ k.transpose(-1, -2) * scale                                         # scale = dim_head^-0.5 [ext]
    if key_bias is not None:                                                    # [B, L] additive, every head / query [ext]
        s = s + key_bias[:, None, None, :]
    return torch.softmax(s, dim=-1) @ v


class CrossAttention(nn.Module):  # unet_struct.txt:17-25, :34-42
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, context=None, key_bias=None):
        ctx = x if context is None else context
        B, N, C = x.shape
        d = C // self.heads
        q = self.to_q(x).view(B, N, self.heads, d).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, self.heads, d).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, self.heads, d).transpose(1, 2)
        if ATTN_RECOMPUTE and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            o = checkpoint(_attention_core, q, k, v, key_bias, d ** -0.5, use_reentrant=False)
        else:
            o = _attention_core(q, k, v, key_bias, d ** -0.5)
        o = o.transpose(1, 2).reshape(B, N, C)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):  # unet_struct.txt:28-30
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        hidden, gate = self.proj(x).chunk(2, dim=-1)                            # [ext] hidden first, gate second
        return hidden * F.gelu(gate)


class FeedForward(nn.Module):  # unet_struct.txt:26-33
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):  # unet_struct.txt:16-47 (module order attn1, ff, attn2, norm1-3)
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.attn1 = CrossAttention(dim, dim, heads)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, ctx_dim, heads)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, context):
        context, key_bias = context if isinstance(context, tuple) else (context, None)
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context, key_bias) + x                    # the mask applies to cross-attention only [ext]
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):  # unet_struct.txt:12-50
    def __init__(self, dim, ctx_dim, heads, groups, depth=1, linear_proj=False):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, ctx_dim, heads) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)
        self.linear_proj = linear_proj

    def forward(self, x, context):
        B, C, H, W = x.shape
        if self.linear_proj:                       # use_linear_projection [ext]: tokens first, then nn.Linear
            h = self.proj_in(self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if self.linear_proj:
            return self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2) + x
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h) + x


def _transformer(c, cfg, level):
    return Transformer2DModel(c, cfg["cross_attention_dim"], per_block(cfg["num_attention_heads"], level), cfg["norm_num_groups"],
                              per_block(cfg["transformer_layers_per_block"], level), cfg["use_linear_projection"])


class Downsample2D(nn.Module):  # unet_struct.txt:111-114
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):  # unet_struct.txt:390-393; nearest 2x then conv [ext]
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    """CrossAttnDownBlock2D / DownBlock2D. Returns (hidden, skip tuple) — reference controlnet.py:149-171."""

    def __init__(self, cin, cout, temb_dim, n_layers, cfg, has_attn, add_down, level=0):
        super().__init__()
        g = cfg["norm_num_groups"]
        if has_attn:
            self.attentions = nn.ModuleList([_transformer(cout, cfg, level) for _ in range(n_layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, g) for i in range(n_layers)])
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_attn = has_attn

    def forward(self, h, temb, context):
        skips = ()
        for i, res in enumerate(self.resnets):
            h = res(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, context)
            skips += (h,)
        if hasattr(self, "downsamplers"):
            h = self.downsamplers[0](h)
            skips += (h,)
        return h, skips


class MidBlock(nn.Module):  # unet_struct.txt:866-928
    def __init__(self, c, temb_dim, cfg):
        super().__init__()
        g = cfg["norm_num_groups"]
        self.attentions = nn.ModuleList([_transformer(c, cfg, len(cfg["block_out_channels"]) - 1)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, g), ResnetBlock2D(c, c, temb_dim, g)])

    def forward(self, h, temb, context):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    """UpBlock2D / CrossAttnUpBlock2D: concat order is [hidden, skip] (reference controlnet.py:73-75)."""

    def __init__(self, cin, cout, prev, temb_dim, n_layers, cfg, has_attn, add_up, level=0):
        super().__init__()
        g = cfg["norm_num_groups"]
        if has_attn:
            self.attentions = nn.ModuleList([_transformer(cout, cfg, level) for _ in range(n_layers)])
        res = []
        for i in range(n_layers):
            skip_c = cin if i == n_layers - 1 else cout
            in_c = prev if i == 0 else cout
            res.append(ResnetBlock2D(in_c + skip_c, cout, temb_dim, g))
        self.resnets = nn.ModuleList(res)
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_attn = has_attn

    def forward(self, h, skips, temb, context):
        for i, res in enumerate(self.resnets):
            h = torch.cat([h, skips[-1]], dim=1)
            skips = skips[:-1]
            h = res(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, context)
        if hasattr(self, "upsamplers"):
            h = self.upsamplers[0](h)
        return h


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class OracleUNet2DConditionModel(nn.Module):
    """fp32 reference of diffusers UNet2DConditionModel (SD1.x family). Call signature: wrapper.py:29."""

    def __init__(self, **cfg):
        super().__init__()
        cfg = dict(SD15_CONFIG, **cfg)
        self.config = cfg
        boc = cfg["block_out_channels"]
        temb_dim = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, 1, 1)
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim)
        n = cfg["layers_per_block"]
        downs, out_c = [], boc[0]
        for i, t in enumerate(cfg["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            downs.append(DownBlock(in_c, out_c, temb_dim, n, cfg, t.startswith("CrossAttn"), i != len(boc) - 1, level=i))
        self.down_blocks = nn.ModuleList(downs)
        ups, rev = [], list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(in_c, out_c, prev, temb_dim, n + 1, cfg, t.startswith("CrossAttn"), i != len(boc) - 1,
                               level=len(boc) - 1 - i))          # reversed heads / transformer depth [ext]
        self.up_blocks = nn.ModuleList(ups)
        self.mid_block = MidBlock(boc[-1], temb_dim, cfg)
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, 1, 1)
        if cfg["addition_embed_type"] == "text_time":            # SDXL micro-conditioning [ext]
            self.add_time_proj = Timesteps(cfg["addition_time_embed_dim"])
            self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], temb_dim)
        else:
            assert cfg["addition_embed_type"] is None

    def forward(self, sample, timestep, encoder_hidden_states, encoder_attention_mask=None, added_cond_kwargs=None,
                control_residuals=None, **kwargs):
        """control_residuals: the 13 tensors of a ControlNet branch, applied as the reference's to_layer_hook does
        (hcpdiff/models/controlnet.py:70-82): skip[j] += r[j] (r[0] on the conv_in skip), mid output += r[-1]."""
        if encoder_attention_mask is not None:     # diffusers UNet2DConditionModel.forward [ext]: (1 - mask) * -10000, unsqueeze(1)
            encoder_hidden_states = (encoder_hidden_states, (1.0 - encoder_attention_mask.to(sample.dtype)) * -10000.0)
        temb = self.time_embedding(self.time_proj(timestep).to(sample.dtype))
        if self.config["addition_embed_type"] == "text_time":    # call contract: reference wrapper.py:66,73
            text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            time_embeds = self.add_time_proj(time_ids.flatten()).reshape(text_embeds.shape[0], -1)
            add_embeds = torch.cat([text_embeds, time_embeds], dim=-1).to(temb.dtype)
            temb = temb + self.add_embedding(add_embeds)
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, s = blk(h, temb, encoder_hidden_states)
            skips += s
        h = self.mid_block(h, temb, encoder_hidden_states)
        if control_residuals is not None:
            assert len(control_residuals) == len(skips) + 1
            skips = tuple(s + r for s, r in zip(skips, control_residuals[:-1]))
            h = h + control_residuals[-1]
        for blk in self.up_blocks:
            k = len(blk.resnets)
            h = blk(h, skips[-k:], temb, encoder_hidden_states)
            skips = skips[:-k]
        return UNetOutput(self.conv_out(self.conv_act(self.conv_norm_out(h))))


class OracleControlNet(nn.Module):
    """fp32 restatement of the reference ControlNetPlugin (hcpdiff/models/controlnet.py:11-62 construction, :88-183
    forward): deep copy of the host encoder + cond_head conv/SiLU stack + zero 1x1 convs; same parameter names."""

    def __init__(self, host, cond_block_channels=None, layers_per_block=None, block_out_channels=None):
        super().__init__()
        from copy import deepcopy
        boc = tuple(block_out_channels or host.config["block_out_channels"])
        n = layers_per_block if layers_per_block is not None else host.config["layers_per_block"]
        ch = tuple(cond_block_channels or (3, 16, 32, 96, 256, boc[0]))
        self.conv_in = deepcopy(host.conv_in)
        self.time_proj = deepcopy(host.time_proj)
        self.time_embedding = deepcopy(host.time_embedding)
        self.down_blocks = deepcopy(host.down_blocks)
        self.mid_block = deepcopy(host.mid_block)
        head = [nn.Conv2d(ch[0], ch[1], 3, padding=1), nn.SiLU()]                       # controlnet.py:46-56
        for i in range(2, (len(ch) - 2) * 2):
            head += [nn.Conv2d(ch[i // 2], ch[(i + 1) // 2], 3, padding=1, stride=1 + i % 2), nn.SiLU()]
        head.append(nn.Conv2d(ch[-2], ch[-1], 3, padding=1))
        self.cond_head = nn.Sequential(*head)
        zero = [nn.Conv2d(boc[0], boc[0], 1)] + [nn.Conv2d(c, c, 1) for c in boc for _ in range(n + 1)]   # controlnet.py:30-35
        self.controlnet_mid_block = zero.pop()
        self.controlnet_down_blocks = nn.ModuleList(zero)
        for m in list(self.controlnet_down_blocks) + [self.controlnet_mid_block, self.cond_head[-1]]:
            nn.init.constant_(m.weight, 0)                                              # controlnet.py:58-62

    def forward(self, sample, timestep, encoder_hidden_states, cond):
        temb = self.time_embedding(self.time_proj(timestep).to(sample.dtype))
        h = self.conv_in(sample) + self.cond_head(cond)                                 # controlnet.py:143-147
        res = (h,)
        for blk in self.down_blocks:
            h, s = blk(h, temb, encoder_hidden_states)
            res += s
        h = self.mid_block(h, temb, encoder_hidden_states)
        out = tuple(zc(r) for r, zc in zip(res, self.controlnet_down_blocks))           # controlnet.py:174-181
        return out + (self.controlnet_mid_block(h),)


def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDPMScheduler(beta_schedule='scaled_linear') [ext]; same constants restated in the reference at
    hcpdiff/loggers/preview/image_previewer.py:28."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0, noise, timesteps, alphas_cumprod):
    """DDPMScheduler.add_noise [ext] as called by reference train_ac.py:447."""
    a = alphas_cumprod[timesteps].to(x0.dtype)
    shape = (-1,) + (1,) * (x0.dim() - 1)
    return a.sqrt().view(shape) * x0 + (1 - a).sqrt().view(shape) * noise


def seeded_init_(model, seed=0):
    """Deterministic non-degenerate weights (no pretrained weights exist on this box).  Every parameter gets its own
    CPU generator seeded from (seed, crc32(name)), so any model with diffusers parameter names — the oracle, the native
    UNet, on any device — receives bit-identical values regardless of module registration order.  Matrices ~ N(0, 1/fan_in);
    norm scales ~ 1 + 0.1 N(0,1); biases / norm shifts ~ 0.05 N(0,1).  LoRA factors are left untouched."""
    import zlib
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "lora_block_" in name:
                continue
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.replace("._host", "").encode())) % (2 ** 31))
            if p.dim() == 1:
                if "norm" in name and name.endswith("weight"):
                    v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
                else:
                    v = 0.05 * torch.randn(p.shape, generator=g)
            else:
                v = torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel())
            p.copy_(v)
    return model
