"""ORACLE / test infrastructure (never imported by the product path): CPU fp32 restatement of the encoder half of
``AutoencoderKL`` — what the reference calls as ``vae.encode(image).latent_dist.sample() * vae.config.scaling_factor``
(/root/reference/hcpdiff/data/pair_dataset.py:72-75 for the latent cache, hcpdiff/train_ac.py:428-435 on line; the VAE is
loaded at train_ac.py:212-215 from the un-vendored ``diffusers<=0.26.1``).

**Parity unpinned**: the arithmetic lives in diffusers (absent here, no golden vectors in the reference), so this file
restates the public diffusers 0.26 architecture — every statement below is [ext] knowledge, unverifiable offline:

  Encoder: conv_in 3->128 (3x3, pad 1); 4 DownEncoderBlock2D over block_out_channels (128, 256, 512, 512), layers_per_block=2
    ResnetBlock2D(temb=None, groups 32, eps 1e-6): GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + x (1x1 conv_shortcut when
    the channel count changes), output_scale_factor 1;
    Downsample2D(padding=0) after blocks 0-2: F.pad(x, (0,1,0,1)) then conv3x3 stride 2 padding 0;
  mid_block: ResnetBlock2D, Attention(heads=1, dim_head=512, group_norm(32, eps 1e-6), bias, residual_connection=True), ResnetBlock2D;
  conv_norm_out GN(32, eps 1e-6) -> SiLU -> conv_out 512 -> 2*latent (3x3, pad 1);
  quant_conv 1x1 (2*latent -> 2*latent); DiagonalGaussianDistribution: mean, logvar = chunk(2, dim=1), logvar clamp [-30, 20],
  sample = mean + exp(0.5 logvar) * randn.  scaling_factor 0.18215 (SD1.x/2.x), 0.13025 (SDXL).
Parameter names are diffusers' (``encoder.down_blocks.0.resnets.0.norm1.weight`` ... ``quant_conv.bias``) so that a diffusers
``vae/diffusion_pytorch_model.safetensors`` loads with strict=False (the decoder keys are ignored).

Round 4 — the DECODER half, ``vae.decode(latents / vae.config.scaling_factor)`` as the reference's preview / inference pipeline
calls it (/root/reference/hcpdiff/utils/pipe_hook.py:154-155 under loggers/preview/image_previewer.py:97-149 and
workflow/diffusion.py; same [ext] status):
  post_quant_conv 1x1 (latent -> latent); Decoder: conv_in latent -> 512 (3x3, pad 1); mid_block as above; 4 UpDecoderBlock2D over the
  REVERSED block_out_channels (512, 512, 256, 128), layers_per_block + 1 = 3 ResnetBlock2D each (the first takes the previous block's
  width), Upsample2D after blocks 0-2: nearest-neighbour x2 then conv3x3 pad 1; conv_norm_out GN(32, eps 1e-6) -> SiLU -> conv_out
  128 -> 3 (3x3, pad 1).  Names: ``decoder.up_blocks.0.resnets.0.norm1.weight`` ... ``post_quant_conv.bias``.
"""
import torch
import torch.nn.functional as F
from torch import nn

SD_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                     scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8,
                       scaling_factor=0.18215)


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return h + (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x)


class VaeDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class VaeDownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        if down:
            self.downsamplers = nn.ModuleList([VaeDownsample(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0](x)
        return x


class VaeAttention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c); self.to_k = nn.Linear(c, c); self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)                # [B, HW, C], one head of width C
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), -1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class VaeMidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])
        self.resnets = nn.ModuleList([VaeResnet(c, c, groups), VaeResnet(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class VaeEncoderNet(nn.Module):
    def __init__(self, in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups):
        super().__init__()
        boc = block_out_channels
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([VaeDownBlock(boc[max(i - 1, 0)], boc[i], layers_per_block, norm_num_groups, i < len(boc) - 1)
                                          for i in range(len(boc))])
        self.mid_block = VaeMidBlock(boc[-1], norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x = blk(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class OracleVAEEncoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                 scaling_factor=0.18215):
        super().__init__()
        self.encoder = VaeEncoderNet(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.scaling_factor = scaling_factor

    def moments(self, image):
        return self.quant_conv(self.encoder(image))

    def encode(self, image, noise=None):
        """vae.encode(image).latent_dist.sample() * scaling_factor with the draw's noise given explicitly (None: the mode)."""
        mean, logvar = self.moments(image).chunk(2, 1)
        logvar = logvar.clamp(-30.0, 20.0)
        z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
        return z * self.scaling_factor


class VaeUpsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class VaeUpBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        if up:
            self.upsamplers = nn.ModuleList([VaeUpsample(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0](x)
        return x


class VaeDecoderNet(nn.Module):
    def __init__(self, out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups):
        super().__init__()
        rev = tuple(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0], norm_num_groups)
        self.up_blocks = nn.ModuleList([VaeUpBlock(rev[max(i - 1, 0)], rev[i], layers_per_block + 1, norm_num_groups, i < len(rev) - 1)
                                        for i in range(len(rev))])
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for blk in self.up_blocks:
            x = blk(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class OracleAutoencoderKL(OracleVAEEncoder):
    """Encoder + decoder: ``decode(z)`` = decoder(post_quant_conv(z)) on UNSCALED latents (the caller divides by scaling_factor,
    pipe_hook.py:155)."""

    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                 scaling_factor=0.18215):
        super().__init__(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups, scaling_factor)
        self.decoder = VaeDecoderNet(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
