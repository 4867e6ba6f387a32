"""Leaf modules of the native UNet.  They ARE ``nn.Linear`` / ``nn.Conv2d`` / ``nn.GroupNorm`` / ``nn.LayerNorm``
subclasses with diffusers parameter names and shapes, so the reference's plugin machinery keeps working:
``isinstance(layer, (nn.Linear, nn.Conv2d))`` LoRA wrapping (reference lora_base_patch.py:39, plugin.py:308),
``state_dict`` / checkpoint names (ckpt_manager), ``deepcopy`` (controlnet.py:38-44).

fp32 parameters are the masters; each leaf lazily keeps bf16 operand copies in the layouts the kernels consume
(re-packed when the parameter's version counter changes).  Activations: bf16, channels-last.
"""
import torch
from torch import nn

from . import ops

BF16 = torch.bfloat16


class _Packed:
    pass


def _key(*ts):
    return tuple((t._version, t.data_ptr(), t.device) if t is not None else None for t in ts)


def _same_storage(k_old, k_new):
    """Only the version counters moved (an optimizer stepped the parameter in place): the bf16 operand buffers are refreshed IN
    PLACE then, so that captured hipGraphs (graphed.py) and grouped pack descriptors (fullft.py) that hold their addresses stay valid."""
    return len(k_old) == len(k_new) and all((a is None) == (b is None) and (a is None or a[1:] == b[1:]) for a, b in zip(k_old, k_new))


def _pad_last(t, mult):
    c = t.shape[-1]
    cp = (c + mult - 1) // mult * mult
    if cp == c:
        return t
    out = t.new_zeros(t.shape[:-1] + (cp,))
    out[..., :c] = t
    return out


class HipLinear(nn.Linear):
    supports_fused_residual = True
    _pk = None

    def packed(self):
        k = _key(self.weight, self.bias)
        if self._pk is None or self._pk.key != k:
            w = self.weight.detach()
            if self._pk is not None and _same_storage(self._pk.key, k):
                pk = self._pk
                pk.w.copy_(w); pk.wt.copy_(w.t())
                if pk.bias is not None and pk.bias.data_ptr() != self.bias.data_ptr():
                    pk.bias.copy_(self.bias.detach())
                pk.key = k
                return pk
            pk = _Packed(); pk.key = k
            pk.w = w.to(BF16).contiguous()                 # [N, K]   forward B operand
            pk.wt = w.t().to(BF16).contiguous()            # [K, N]   dX B operand
            pk.bias = self.bias.detach().float().contiguous() if self.bias is not None else None
            self._pk = pk
        return self._pk

    def forward(self, x, residual=None, out_f32=False):
        return ops.linear(x, self, None, residual, out_f32)


class HipConv2d(nn.Conv2d):
    """1x1 (a GEMM on channels-last tokens) or 3x3/pad-1 convolution over NHWC bf16 activations."""
    supports_fused_residual = True
    _pk = None

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.kernel_size == (3, 3):
            # diffusers shape [Cout,Cin,3,3], channels_last STORAGE = the kernels' [Cout][ky][kx][Cin], from construction on: the flat
            # fine-tuning bucket (fullft.HostBucket) keeps this layout, so a parameter's strides never change under a wrapper that
            # recorded them when it was built (torch DDP's bucket views: a later switch scrambled the 3x3 gradients it copied back)
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def packed(self):
        k = _key(self.weight, self.bias)
        if self._pk is None or self._pk.key != k:
            w = self.weight.detach()
            if self._pk is not None and _same_storage(self._pk.key, k):
                pk = self._pk
                if self.kernel_size == (1, 1):
                    w2 = w.reshape(pk.cout, pk.cin)
                    pk.w.copy_(w2); pk.wt.copy_(w2.t())
                else:
                    pk.w[..., :pk.cin].copy_(w.permute(0, 2, 3, 1)); pk.wd[..., :pk.cout].copy_(w.permute(1, 2, 3, 0))
                if pk.bias is not None and pk.bias.data_ptr() != self.bias.data_ptr():
                    pk.bias.copy_(self.bias.detach())
                pk.key = k
                return pk
            pk = _Packed(); pk.key = k
            cout, cin = w.shape[0], w.shape[1]
            pk.cout, pk.cin = cout, cin
            pk.bias = self.bias.detach().float().contiguous() if self.bias is not None else None
            if self.kernel_size == (1, 1):
                w2 = w.reshape(cout, cin)
                pk.w = w2.to(BF16).contiguous(); pk.wt = w2.t().to(BF16).contiguous()
            else:
                # 3x3 / padding 1 everywhere in the UNet; padding 0 + stride 2 is the VAE encoder's Downsample2D (vae.py passes pad=0)
                assert self.kernel_size == (3, 3) and (self.padding == (1, 1) or (self.padding == (0, 0) and self.stride == (2, 2))), \
                    "only 3x3 (padding 1, or the VAE's padding-0 stride-2 downsample) and 1x1 convolutions exist in SD models"
                pk.cin_pad = (cin + 7) // 8 * 8
                pk.cout_pad = (cout + 7) // 8 * 8
                pk.w = _pad_last(w.permute(0, 2, 3, 1), 8).to(BF16).contiguous()       # [Cout][ky][kx][Cin_pad]
                pk.wd = _pad_last(w.permute(1, 2, 3, 0), 8).to(BF16).contiguous()      # [Cin][ky][kx][Cout_pad]
            self._pk = pk
        return self._pk

    def forward(self, x, residual=None, x2=None, rowbias=None, upsample=False):
        if self.kernel_size == (1, 1):
            assert x2 is None and rowbias is None and not upsample
            return ops.linear(x, self, None, residual)
        return ops.conv3x3(x, self, x2=x2, rowbias=rowbias, residual=residual, stride=self.stride[0], upsample=upsample)


class HipConvIn(HipConv2d):
    """conv_in: NCHW fp32/bf16 latents -> channel-padded NHWC bf16 -> 3x3 conv.  A real module call, so forward
    (pre-)hooks fire (the reference's ControlNet plugin hooks `pre_hook:conv_in`, cfgs/plugins/plugin_controlnet.yaml)."""

    def forward(self, sample):
        return ops.conv_in(sample, self)


class HipConvOut(HipConv2d):
    """conv_out: NHWC bf16 -> 3x3 conv -> NCHW fp32 (the `.sample` tensor)."""

    def forward(self, x):
        return ops.conv_out(x, self)


class _F32Affine:
    _af = None

    def f32_params(self):
        k = _key(self.weight, self.bias)
        if self._af is None or self._af[0] != k:
            if self._af is not None and _same_storage(self._af[0], k):       # (fp32 contiguous parameters: these ARE the parameters)
                for dst, src in ((self._af[1], self.weight), (self._af[2], self.bias)):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src.detach())
                self._af = (k, self._af[1], self._af[2])
            else:
                self._af = (k, self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous())
        return self._af[1], self._af[2]


class HipGroupNorm(_F32Affine, nn.GroupNorm):
    def forward(self, x, silu=False, fork=False):
        """fork=True returns (norm(x), x): use the second value as the residual so backward fuses the gradient add."""
        return ops.groupnorm_fork(x, self, silu) if fork else ops.groupnorm(x, self, silu)


class HipLayerNorm(_F32Affine, nn.LayerNorm):
    def forward(self, x, fork=False):
        return ops.layernorm_fork(x, self) if fork else ops.layernorm(x, self)
