"""ControlNet branch on the native UNet — the MI355X counterpart of the reference's ``ControlNetPlugin``
(hcpdiff/models/controlnet.py:11-183, config cfgs/plugins/plugin_controlnet.yaml; SURVEY.md §8 a10 / BASELINE configs[4]).

Same role, constructor surface and parameter names as the reference class (``conv_in``, ``time_embedding``,
``down_blocks``, ``mid_block``, ``cond_head.N``, ``controlnet_down_blocks.N``, ``controlnet_mid_block``), so plugin
checkpoints interchange; same hook protocol:

  from_layers  'pre_hook:'          root pre-hook    -> remember (sample, timestep, encoder_hidden_states)   (controlnet.py:64-66)
               'pre_hook:conv_in'   conv_in pre-hook -> run the branch, keep the 13 residuals                (controlnet.py:67-68)
  to_layers    'down_blocks.0-3'    post-hook        -> skip[j] += residual                                  (controlnet.py:77-82)
               'mid_block'          post-hook        -> hidden += residual[12]                               (controlnet.py:79-80)
               'pre_hook:up_blocks.3.resnets.2'      -> the conv_in skip, consumed there, += residual[0]     (controlnet.py:71-76)

The branch is a deep copy of the host's NATIVE encoder, so every op is the same gfx950 kernel; activations stay bf16
channels-last, the residual adds are the fused-add kernel, and with ``train_cfg`` / ``HostBucket`` the branch's 361 M
parameters train through the TN-GEMM weight-gradient kernels.  The control image arrives through the reference's input
feeder protocol (``plugin_input={'cond': [B,3,H,W]}`` -> ``host.input_feeder``; wrapper.py:26-28, controlnet.py:84-86).
"""
import weakref
from copy import deepcopy

import torch
from torch import nn

from . import kernels as K
from . import ops
from .layers import HipConv2d
from .patch_api import BasePluginBlock, MultiPluginBlock, PatchPluginContainer

BF16 = torch.bfloat16


def parse_hook_layers(patterns, named_modules):
    """'pre_hook:<module path>' selectors -> [{'layer': module, 'pre_hook': bool}]  (cfg_net_tools.get_match_layers with
    return_metas=True, as used by make_plugin for MultiPluginBlock, cfg_net_tools.py:148-152)."""
    out = []
    for pat in patterns:
        metas = pat.split(":")
        out.append({"layer": named_modules[metas[-1]], "pre_hook": "pre_hook" in metas[:-1]})
    return out


class _CondHead(nn.Sequential):
    """conv3x3 / SiLU stack on the control image: NCHW fp32 in [0,1] -> channel-padded NHWC bf16 -> ... -> [B,h,w,320]."""

    def forward(self, cond):
        x = K.nchw_to_nhwc(cond.float().contiguous(), 8)
        for m in self:
            x = m(x)
        return x


class _SiLU(nn.Module):
    def forward(self, x):
        return ops.silu(x)


class ControlNetHipPlugin(MultiPluginBlock):
    def __init__(self, name, from_layers, to_layers, host_model=None, cond_block_channels=(3, 16, 32, 96, 256, 320),
                 layers_per_block=2, block_out_channels=(320, 640, 1280, 1280)):
        # a MultiPluginBlock by ROLE (make_plugin's issubclass dispatch, cfg_net_tools.py:148): its own __init__ would register the
        # reference's hook lambdas (plugin.py:189-201); this class registers kwargs-aware hooks below instead
        BasePluginBlock.__init__(self, name)
        assert host_model is not None
        assert len(from_layers) == 2 and len(to_layers) == len(block_out_channels) + 2, "ControlNet hook layout: see plugin_controlnet.yaml"
        self.host_model = weakref.ref(host_model)
        # reachable as host.<name> like the reference (plugin.py:182), but NOT a registered child: the host's
        # requires_grad_/modules()/state_dict stay the host's own (the trainer owns the plugin's parameters)
        object.__setattr__(host_model, name, self)
        if not hasattr(host_model, "input_feeder"):
            host_model.input_feeder = []
        host_model.input_feeder.append(self.feed_input_data)

        self.conv_in = self.copy_block(host_model.conv_in)
        self.time_proj = self.copy_block(host_model.time_proj)
        self.time_embedding = self.copy_block(host_model.time_embedding)
        self.down_blocks = self.copy_block(host_model.down_blocks)
        self.mid_block = self.copy_block(host_model.mid_block)
        # An SDXL host (addition_embed_type 'text_time') is copied the same way: the reference branch takes (sample, timestep,
        # encoder_hidden_states) from the root pre-hook and never sees added_cond_kwargs, nor does it copy add_embedding
        # (controlnet.py:19-25,88-97) — its time embedding carries no text_time term, and neither does this one.

        self.build_head(cond_block_channels)
        boc = block_out_channels
        zero = [HipConv2d(boc[0], boc[0], 1)] + [HipConv2d(c, c, 1) for c in boc for _ in range(layers_per_block + 1)]
        self.controlnet_mid_block = zero.pop()
        self.controlnet_down_blocks = nn.ModuleList(zero)
        self.reset_parameters()
        self.to(next(host_model.parameters()).device)

        self.n_down = len(block_out_channels)
        self._handles = []
        for idx, layer in enumerate(from_layers):
            assert layer["pre_hook"], "ControlNet from_layers are pre-hooks (plugin_controlnet.yaml)"
            self._handles.append(layer["layer"].register_forward_pre_hook(
                lambda host, args, kwargs, idx=idx: self.from_layer_hook(host, args, kwargs, idx), with_kwargs=True))
        for idx, layer in enumerate(to_layers):
            if layer["pre_hook"]:
                self._handles.append(layer["layer"].register_forward_pre_hook(
                    lambda host, args, kwargs, idx=idx: self.to_layer_pre_hook(host, args, kwargs, idx), with_kwargs=True))
            else:
                self._handles.append(layer["layer"].register_forward_hook(
                    lambda host, args, out, idx=idx: self.to_layer_hook(host, args, out, idx)))
        self.cond = None
        self.feat_to = None

    # ---- construction (controlnet.py:38-62)
    @staticmethod
    def copy_block(block):
        if block is None:
            return None
        block = deepcopy(block)
        # remove_all_hooks + remove_layers(block, BasePluginBlock) (controlnet.py:38-44): the branch is a copy of the PLAIN host
        # layers.  (The reference's remove_layers indexes a dict with a list, net_utils.py:141, so there a LoRA-wrapped host ends
        # in a TypeError; its evident intent — drop the plugin blocks, keep the hosts — is what happens here: every patch
        # container in the copy is replaced by the host layer it wraps.)
        if isinstance(block, PatchPluginContainer):
            block = block._host
        for _ in range(8):                                  # containers never nest deeper than this
            swaps = [(parent, name, child._host) for parent in block.modules() for name, child in parent.named_children()
                     if isinstance(child, PatchPluginContainer)]
            if not swaps:
                break
            for parent, name, host in swaps:
                setattr(parent, name, host)
        for m in block.modules():
            m._forward_hooks.clear(); m._forward_pre_hooks.clear(); m._backward_hooks.clear()
            if hasattr(m, "_groups"):
                m._groups = {}
        return block

    def build_head(self, ch):
        head = [HipConv2d(ch[0], ch[1], 3, 1, 1), _SiLU()]
        for i in range(2, (len(ch) - 2) * 2):
            head += [HipConv2d(ch[i // 2], ch[(i + 1) // 2], 3, 1 + i % 2, 1), _SiLU()]
        head.append(HipConv2d(ch[-2], ch[-1], 3, 1, 1))
        self.cond_head = _CondHead(*head)

    def reset_parameters(self):
        for m in list(self.controlnet_down_blocks) + [self.controlnet_mid_block, self.cond_head[-1]]:
            nn.init.constant_(m.weight, 0)                           # zero convs (biases keep their default init, as in the reference)

    # ---- input feeder + hooks
    def feed_input_data(self, data):
        if isinstance(data, dict):
            self.cond = data["cond"]

    def from_layer_hook(self, host, args, kwargs, idx):
        if idx == 0:
            self.data_input = args[:3]
        elif idx == 1:
            self.feat_to = self(*self.data_input)

    def to_layer_hook(self, host, args, out, idx):
        if idx == self.n_down:                                       # mid block
            return ops.add(out, self.feat_to[-1])
        h, skips = out                                               # down block: (hidden, skip tuple)
        base = 1 + sum(len(self.down_blocks[i].resnets) + (1 if hasattr(self.down_blocks[i], "downsamplers") else 0) for i in range(idx))
        return h, tuple(ops.add(s, self.feat_to[base + i]) for i, s in enumerate(skips))

    def to_layer_pre_hook(self, host, args, kwargs, idx):
        """Last resnet of the last up block consumes the conv_in skip: add residual 0 there (controlnet.py:71-76 adds it
        to the second channel half of the concatenated input; the native block takes the skip as its own argument)."""
        kwargs = dict(kwargs)
        kwargs["skip"] = ops.add(kwargs["skip"], self.feat_to[0])
        return args, kwargs

    def remove(self):
        for h in self._handles:
            h.remove()
        host = self.host_model()
        if host is not None:
            host.input_feeder.remove(self.feed_input_data)

    # ---- the branch (controlnet.py:88-183)
    def forward(self, sample, timestep, encoder_hidden_states):
        B = sample.shape[0]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64, device=sample.device)
        timestep = timestep.to(torch.int64).reshape(-1).expand(B).contiguous()
        temb_act = ops.silu(self.time_embedding(self.time_proj(timestep)))
        ctx = encoder_hidden_states
        if ctx.dtype != BF16:
            ctx = ctx.to(BF16)
        ctx = ctx.contiguous()
        if self.cond is None:
            raise RuntimeError("ControlNetHipPlugin: no control image fed (plugin_input={'cond': ...} -> host.input_feeder)")
        h = ops.add(self.conv_in(sample), self.cond_head(self.cond))
        res = (h,)
        for blk in self.down_blocks:
            h, s = blk(h, temb_act, ctx)
            res += s
        h = self.mid_block(h, temb_act, ctx)
        out = tuple(zc(r) for r, zc in zip(res, self.controlnet_down_blocks))
        return out + (self.controlnet_mid_block(h),)


def make_controlnet(unet, name="controlnet1", from_layers=("pre_hook:", "pre_hook:conv_in"), to_layers=None, **kwargs):
    """make_plugin's MultiPluginBlock branch (cfg_net_tools.py:148-162).  Default layer lists = plugin_controlnet.yaml
    (for SD1.5: down_blocks.0-3, mid_block, pre_hook:up_blocks.3.resnets.2 — the resnet that consumes the conv_in skip)."""
    named = dict(unet.named_modules())
    boc = tuple(unet.config["block_out_channels"])
    if to_layers is None:
        to_layers = [f"down_blocks.{i}" for i in range(len(boc))] + [
            "mid_block", f"pre_hook:up_blocks.{len(boc) - 1}.resnets.{unet.config['layers_per_block']}"]
    kwargs.setdefault("block_out_channels", boc)
    kwargs.setdefault("layers_per_block", unet.config["layers_per_block"])
    if "cond_block_channels" not in kwargs:
        kwargs["cond_block_channels"] = (3, 16, 32, 96, 256, boc[0])
    return ControlNetHipPlugin(name, parse_hook_layers(from_layers, named), parse_hook_layers(to_layers, named), host_model=unet, **kwargs)
