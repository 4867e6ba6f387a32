"""Native VAE encoder: images -> the latents the training path consumes.

What the reference runs as ``vae.encode(image).latent_dist.sample() * vae.config.scaling_factor`` — once per image when it
builds the latent cache (hcpdiff/data/pair_dataset.py:60-79) or every step when ``cache_latents`` is off
(hcpdiff/train_ac.py:428-435) — on diffusers' ``AutoencoderKL`` (un-vendored; the public 0.26 architecture is restated in
oracle/vae_ref.py).  Inference only: the VAE is frozen in every reference config (train_ac.py:264-266).

Same gfx950 kernels as the UNet: implicit-GEMM 3x3 convolutions over channels-last bf16 (``pad=0`` for the encoder's asymmetric
stride-2 Downsample2D), GroupNorm+SiLU, GEMM linears.  The mid block's single 512-wide attention head does not fit the
flash kernels (head dims 40-160): it runs as S = Q K^T (fp32 out) -> row softmax -> P V^T^T, two GEMMs and two small kernels per
sample; S is 64 MB at 512 px — nothing next to 288 GB.  ``quant_conv``, the logvar clamp and the reparameterised draw are one
kernel over the fp32 moments.  Parameter names and shapes are diffusers' (``encoder.*``, ``quant_conv.*``): a diffusers
``vae/`` directory loads directly (decoder keys ignored).

The reference keeps the VAE in fp32 (``vae_dtype``, train_ac.py:274); here activations are bf16 with fp32 accumulation and
fp32 statistics, like the UNet — latents agree with the fp32 oracle to bf16 rounding (tests/test_vae.py states the tolerance).
Image sides up to 1024 px (the convolution kernels index output pixels with 10 bits per axis): SD 512 px and SDXL 1024 px.

Round 4: ``NativeAutoencoderKL`` adds the DECODER half — ``vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]``
as the reference's preview / inference pipeline calls it (hcpdiff/utils/pipe_hook.py:154-155, reached from
loggers/preview/image_previewer.py:97-149 and workflow/diffusion.py's decode action) — from the same kernels: the nearest-2x upsample
rides in the convolution's gather (``upsample=True``), ``post_quant_conv`` (a per-pixel L x L matrix) is folded into ``conv_in``
exactly: the latent goes in as [z | 1 | 0..] channels, the composed weight's "1" column carries W_in * b_pq per tap, so the zero
padding of the border taps drops the bias there just as in the two-op form.
"""
import json
import os

import torch
from torch import nn

from . import kernels as K
from .layers import HipConv2d, HipGroupNorm, HipLinear

BF16 = torch.bfloat16

SD_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                     scaling_factor=0.18215)


class _Config(dict):
    """diffusers configs answer both ``cfg['k']`` and ``cfg.k``; the reference reads ``vae.config.scaling_factor`` (pair_dataset.py:75)."""
    __getattr__ = dict.__getitem__


class _LatentDist:
    """What ``vae.encode(image).latent_dist`` must offer the reference: ``.sample()`` (data/pair_dataset.py:74, train_ac.py:431) and
    ``.mode()``; both are one pass of hcp_vae_latent_sample over the encoder's fp32 moments (quant_conv folded in), UNSCALED as in
    diffusers — the caller multiplies by ``vae.config.scaling_factor``."""

    def __init__(self, vae, moments):
        self._vae, self._moments = vae, moments

    def sample(self, generator=None, noise=None):
        return self._vae._draw(self._moments, generator, noise, True, 1.0)

    def mode(self):
        return self._vae._draw(self._moments, None, None, False, 1.0)


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


def _gn(m, x, silu):
    g, b = m.f32_params()
    return K.groupnorm_fwd(x, g, b, m.num_groups, m.eps, silu)[0]


def _conv(m, x, residual=None, **kw):
    pk = m.packed()
    return K.conv3x3(x, pk.w, pk.cout, bias=pk.bias, residual=residual, **kw)


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, eps=1e-6)
        self.conv1 = HipConv2d(cin, cout, 3, padding=1)
        self.norm2 = HipGroupNorm(groups, cout, eps=1e-6)
        self.conv2 = HipConv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = HipConv2d(cin, cout, 1)

    def forward(self, x):
        h = _conv(self.conv1, _gn(self.norm1, x, True))
        if hasattr(self, "conv_shortcut"):
            pk = self.conv_shortcut.packed()
            x = K.gemm(x.view(-1, x.shape[-1]), pk.w, bias=pk.bias).view(*x.shape[:-1], pk.cout)
        return _conv(self.conv2, _gn(self.norm2, h, True), residual=x)


class VaeDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return _conv(self.conv, x, stride=2, pad=0)           # F.pad(x, (0,1,0,1)) + padding 0, folded into the gather


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
    """One attention head as wide as the feature map's channel count (512 for the SD VAE)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = HipGroupNorm(groups, c, eps=1e-6)
        self.to_q = HipLinear(c, c); self.to_k = HipLinear(c, c); self.to_v = HipLinear(c, c)
        self.to_out = nn.ModuleList([HipLinear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, H, W, C = x.shape
        n = H * W
        h = _gn(self.group_norm, x, False).view(B * n, C)
        q, k, v = (K.gemm(h, m.packed().w, bias=m.packed().bias).view(B, n, C) for m in (self.to_q, self.to_k, self.to_v))
        vt = K.transpose_bf16(v)                                              # [B, C, n]: the [N, K] operand of P V
        o = torch.empty((B, n, C), dtype=BF16, device=x.device)
        for b in range(B):
            s = K.gemm(q[b], k[b], out_f32=True)                             # [n, n] fp32 scores
            K.gemm(K.softmax_rows(s, C ** -0.5), vt[b], out=o[b])
        po = self.to_out[0].packed()
        return K.gemm(o.view(B * n, C), po.w, bias=po.bias, residual=x.view(B * n, C)).view(B, H, W, C)


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
        self.conv_in = HipConv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([VaeDownBlock(boc[max(i - 1, 0)], boc[i], layers_per_block, norm_num_groups, i < len(boc) - 1)
                                          for i in range(len(boc))])
        self.mid_block = VaeMidBlock(boc[-1], norm_num_groups)
        self.conv_norm_out = HipGroupNorm(norm_num_groups, boc[-1], eps=1e-6)
        self.conv_out = HipConv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, image):
        """image NCHW fp32/bf16 -> conv_out moments, NCHW fp32 [B, 2L, h, w]."""
        pk = self.conv_in.packed()
        x = K.conv3x3(K.nchw_to_nhwc(image.contiguous(), pk.cin_pad), pk.w, pk.cout, bias=pk.bias)
        for blk in self.down_blocks:
            x = blk(x)
        x = self.mid_block(x)
        po = self.conv_out.packed()
        y = K.conv3x3(_gn(self.conv_norm_out, x, True), po.w, po.cout, bias=po.bias, out_f32=True)
        return K.nhwc_to_nchw_f32(y, po.cout)


class NativeVAEEncoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                 scaling_factor=0.18215, **unused):
        super().__init__()
        self.config = _Config(in_channels=in_channels, latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.encoder = VaeEncoderNet(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = HipConv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.requires_grad_(False)

    @property
    def device(self):
        return self.quant_conv.weight.device

    @property
    def dtype(self):                                   # train_ac.py:431 casts the image to vae.dtype before encode
        return self.quant_conv.weight.dtype

    def _check(self, image):
        if image.dim() != 4 or image.shape[1] != self.config["in_channels"]:
            raise ValueError(f"expected an image batch [B,{self.config['in_channels']},H,W], got {tuple(image.shape)}")
        if max(image.shape[2:]) > 1024 or image.shape[2] % 8 or image.shape[3] % 8:
            raise NotImplementedError("hcp_diffusion_amd: VAE encode takes image sides that are multiples of 8 up to 1024 px")

    def _draw(self, mom, generator, noise, sample, scale):
        B, L2, h, w = mom.shape
        if sample and noise is None:
            noise = torch.randn((B, L2 // 2, h, w), dtype=torch.float32, device=mom.device, generator=generator)
        wq = self.quant_conv.weight.detach().float().reshape(L2, L2).contiguous()
        return K.vae_latent_sample(mom, wq, self.quant_conv.bias.detach().float().contiguous(), noise if sample else None, scale)

    @torch.no_grad()
    def encode(self, image):
        """The diffusers call contract the reference uses: ``vae.encode(image).latent_dist.sample()`` (unscaled fp32 [B, L, H/8, W/8];
        data/pair_dataset.py:74, train_ac.py:431) — ``PairDataset.cache_latents`` / ``Trainer.get_latents`` run unchanged on this class."""
        self._check(image)
        return _EncoderOutput(_LatentDist(self, self.encoder(image)))

    @torch.no_grad()
    def encode_latents(self, image, generator=None, noise=None, sample=True):
        """``vae.encode(image).latent_dist.sample() * vae.config.scaling_factor`` in one call (the scale rides in the sampling kernel).
        The draw uses ``noise`` if given, else torch.randn with ``generator`` (the reference draws from torch's global generator);
        sample=False returns the distribution's mode."""
        self._check(image)
        return self._draw(self.encoder(image), generator, noise, sample, self.config["scaling_factor"])

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", device="cuda"):
        """A diffusers model directory (``<path>/vae/config.json`` + ``diffusion_pytorch_model.safetensors``): every tensor of THIS class
        by name — encoder and quant_conv here (decoder keys ignored), all four parts for NativeAutoencoderKL."""
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        cfg = json.load(open(os.path.join(root, "config.json")))
        keys = ("in_channels", "latent_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "scaling_factor")
        model = cls(**{k: cfg[k] for k in keys if k in cfg})
        sd = load_file(os.path.join(root, "diffusion_pytorch_model.safetensors"))
        own = model.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise ValueError(f"VAE checkpoint lacks {len(missing)} tensors of {cls.__name__}, e.g. {missing[:3]}")
        model.load_state_dict({k: sd[k] for k in own})
        return model.to(device)


def build_latent_cache(vae, items, cache_path=None, generator=None):
    """PairDataset.cache_latents (data/pair_dataset.py:60-79): ``items`` yields (img_name, image [3,H,W] in [-1,1], mask [H/8,W/8] or
    None); returns / saves ``{img_name: {'img': latents [L,h,w] cpu fp32, 'mask': [h,w]}}`` — the dict the reference torch.load()s
    from ``cache_path`` (:61-63) and torch.save()s to it (:78-79).  Images of one size can be batched by the caller; this helper
    mirrors the reference's one-image-at-a-time loop."""
    if cache_path and os.path.exists(cache_path):
        return torch.load(cache_path)
    latents = {}
    for name, image, mask in items:
        if name in latents:
            continue
        z = vae.encode_latents(image.unsqueeze(0).to(vae.device), generator=generator).squeeze(0)
        if mask is None:
            mask = torch.ones((z.shape[1], z.shape[2]))
        latents[name] = {"img": z.cpu(), "mask": mask}
    if cache_path:
        torch.save(latents, cache_path)
    return latents


class VaeUpsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, padding=1)

    def forward(self, x):
        return _conv(self.conv, x, upsample=True)             # F.interpolate(x, 2.0, "nearest") folded into the gather


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
        self.conv_in = HipConv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0], norm_num_groups)
        self.up_blocks = nn.ModuleList([VaeUpBlock(rev[max(i - 1, 0)], rev[i], layers_per_block + 1, norm_num_groups, i < len(rev) - 1)
                                        for i in range(len(rev))])
        self.conv_norm_out = HipGroupNorm(norm_num_groups, rev[-1], eps=1e-6)
        self.conv_out = HipConv2d(rev[-1], out_channels, 3, padding=1)


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class NativeAutoencoderKL(NativeVAEEncoder):
    """Encoder (training path, see above) + decoder (preview / inference).  ``decode(z)`` takes UNSCALED latents [B, L, h, w] (fp32 or
    bf16; the reference divides by ``vae.config.scaling_factor`` first, pipe_hook.py:155) and returns images [B, 3, 8h, 8w] fp32 in
    the VAE's [-1, 1] range; ``return_dict=False`` -> a 1-tuple, as diffusers."""

    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32,
                 scaling_factor=0.18215, **unused):
        super().__init__(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups, scaling_factor)
        self.decoder = VaeDecoderNet(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.post_quant_conv = HipConv2d(latent_channels, latent_channels, 1)
        self.requires_grad_(False)
        self._dec_in = None
        self._slicing = False

    # diffusers' memory switches the previewer flips (image_previewer.py:79-94): slicing = one sample at a time; tiling has nothing
    # to save next to 288 GB and is accepted as a no-op
    def enable_slicing(self): self._slicing = True
    def disable_slicing(self): self._slicing = False
    def enable_tiling(self): pass
    def disable_tiling(self): pass

    def _decoder_in(self):
        """conv_in composed with post_quant_conv: [C0][3][3][8] bf16 over the channels [z_0..z_{L-1} | 1 | 0..] (+ conv_in's own bias)."""
        ci, pq = self.decoder.conv_in, self.post_quant_conv
        key = (ci.weight.data_ptr(), ci.weight._version, ci.bias.data_ptr(), ci.bias._version, pq.weight.data_ptr(), pq.weight._version,
               pq.bias.data_ptr(), pq.bias._version, str(ci.weight.device))
        if self._dec_in is None or self._dec_in[0] != key:
            L = self.config["latent_channels"]
            if L + 1 > 8:
                raise NotImplementedError("hcp_diffusion_amd: VAE decode folds post_quant_conv for latent_channels <= 7")
            w_in = ci.weight.detach().float()                                    # [C0, L, 3, 3]
            w_pq = pq.weight.detach().float().reshape(L, L); b_pq = pq.bias.detach().float()
            w = torch.zeros((w_in.shape[0], 3, 3, 8), dtype=torch.float32, device=w_in.device)
            w[..., :L] = torch.einsum("omyx,mi->oyxi", w_in, w_pq)               # (pack time, once per weight version)
            w[..., L] = torch.einsum("omyx,m->oyx", w_in, b_pq)
            self._dec_in = (key, w.to(BF16).contiguous(), ci.bias.detach().float().contiguous())
        return self._dec_in[1], self._dec_in[2]

    def _conv_out(self, x):
        """3 output channels: the kernels store 4-column groups, so the packed weight carries one zero row."""
        co = self.decoder.conv_out
        pk = co.packed()
        if getattr(pk, "w4", None) is None or pk.w4_key != pk.key:
            n4 = (pk.cout + 3) // 4 * 4
            w4 = torch.zeros((n4, 3, 3, pk.w.shape[-1]), dtype=BF16, device=pk.w.device); w4[:pk.cout] = pk.w
            b4 = torch.zeros(n4, dtype=torch.float32, device=pk.w.device)
            if pk.bias is not None:
                b4[:pk.cout] = pk.bias
            pk.w4, pk.b4, pk.w4_key = w4, b4, pk.key
        y = K.conv3x3(_gn(self.decoder.conv_norm_out, x, True), pk.w4, pk.w4.shape[0], bias=pk.b4, out_f32=True)
        return K.nhwc_to_nchw_f32(y, pk.cout)

    def _decode(self, z):
        L = self.config["latent_channels"]
        w, b = self._decoder_in()
        x = K.nchw_to_nhwc(z.contiguous(), 8)
        x[..., L] = 1.0                                                          # the bias channel of the folded post_quant_conv
        x = K.conv3x3(x, w, w.shape[0], bias=b)
        x = self.decoder.mid_block(x)
        for blk in self.decoder.up_blocks:
            x = blk(x)
        return self._conv_out(x)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        if z.dim() != 4 or z.shape[1] != self.config["latent_channels"]:
            raise ValueError(f"expected latents [B,{self.config['latent_channels']},h,w], got {tuple(z.shape)}")
        if max(z.shape[2:]) > 128:
            raise NotImplementedError("hcp_diffusion_amd: VAE decode takes latent sides up to 128 (1024 px images)")
        z = z.to(torch.float32) if z.dtype not in (torch.float32, BF16) else z
        if self._slicing and z.shape[0] > 1:
            img = torch.cat([self._decode(z[i:i + 1]) for i in range(z.shape[0])])
        else:
            img = self._decode(z)
        return _DecoderOutput(img) if return_dict else (img,)
