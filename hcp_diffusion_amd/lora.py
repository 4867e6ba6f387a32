"""Native LoRA layer — seam 2 of the drop-in boundary: ``lora_layer_map['lora_hip']``.

Same constructor/classmethod surface, parameter names (``layer.W_down [r,in]``, ``layer.W_up [out,r]``, buffer
``alpha = cfg_alpha / rank``), init (kaiming_uniform(a=sqrt 5) / zeros) and checkpoint keys as the reference's
``LoraLayer`` (hcpdiff/models/lora_layers_patch.py:26-57, lora_base_patch.py:37-173); the arithmetic differs only in
formulation: instead of materialising ``W + alpha * W_up @ W_down`` every forward (lora_base_patch.py:61-74), the
container runs the host GEMM with the low-rank side path folded in as a 32-wide K-extension (csrc/gemm.hip), and the
rank-r weight gradients are accumulated by csrc/lora.hip directly into a flat fp32 bucket.
"""
import math
import struct

import torch
from torch import nn

from . import kernels as K
from . import ops
from .layers import HipConv2d, HipLinear
from .patch_api import PatchPluginBlock, PatchPluginContainer, PluginGroup

BF16 = torch.bfloat16
RANK_SLOT = 32   # the K-extension width every LoRA layer is padded to (one MFMA k-step)


class LoraHipContainer(PatchPluginContainer):
    """Stands where the host Linear / Conv2d stood (reference LoraPatchContainer, lora_base_patch.py:19-35)."""
    supports_fused_residual = True

    _multis = None        # {plugin-name tuple: MultiLora}: the shared operand images of several blocks on this host

    def forward(self, x, residual=None, **kwargs):
        return self._run(tuple(self.plugin_names), x, residual, **kwargs)

    def _run(self, names, x, residual=None, drop_block=None, **kwargs):
        """The host layer with the LoRA blocks `names` (plugin names of this container) applied — all of them for the plain container;
        one branch's for DAPPHipContainer.  drop_block: the block whose dropout acts on the output (default: the last of `names`; the
        DAPP container hands the container's LAST plugin to BOTH halves, lora_layers_patch.py:132-133)."""
        blocks = [self[n] for n in names]
        b0 = blocks[0]
        if b0.merged or (b0.host_type == "conv" and len(blocks) > 1 and
                         (sum(8 * ((b.rank + 7) // 8) for b in blocks) > RANK_SLOT or any(b.wide for b in blocks))):
            # conv_in / conv_out, or stacked blocks on a 3x3 conv whose ranks do not fit the 32 rank slots: the reference's merged-weight
            # form through the host's own kernels (ops._MergedLoraFn) — any number of blocks, any ranks
            last = drop_block if drop_block is not None else blocks[-1]
            drop = last.dropout.p > 0.0 and last.training
            y = ops.merged_lora_call(self._host, blocks, x, residual=None if drop else residual, **kwargs)
            return self._dropped(y, last, residual) if drop else y
        if len(names) != 1:                            # several blocks on one host: their rank slots side by side
            multi = self._multis.get(names) if self._multis else None
            if multi is None:
                if self._multis is None:
                    self._multis = {}
                multi = self._multis[names] = MultiLora(blocks, names)
            last = drop_block if drop_block is not None else blocks[-1]      # the reference applies the LAST block's dropout (lora_base_patch.py:35)
            drop = last.dropout.p > 0.0 and last.training
            if multi.host_type == "conv":              # 3x3 host: T = conv3x3(x, [W_down_0; W_down_1; ...]) fills the shared rank slots
                host = self._host
                y = ops.conv3x3(x, host, x2=kwargs.pop("x2", None), rowbias=kwargs.pop("rowbias", None), residual=None if drop else residual,
                                stride=host.stride[0], upsample=kwargs.pop("upsample", False), lora=multi, **kwargs)
                return self._dropped(y, last, residual) if drop else y
            if kwargs:
                raise NotImplementedError(f"LoraHipContainer: unsupported call arguments {list(kwargs)}")
            if drop:
                return self._dropped(ops.linear(x, self._host, multi, None), last, residual)
            return ops.linear(x, self._host, multi, residual)
        blk = blocks[0]
        dblk = drop_block if drop_block is not None else blk
        drop = dblk.dropout.p > 0.0 and dblk.training
        if blk.host_type == "conv":                    # 3x3 host: same keyword surface as HipConv2d.forward
            host = self._host
            y = ops.conv3x3(x, host, x2=kwargs.pop("x2", None), rowbias=kwargs.pop("rowbias", None), residual=None if drop else residual,
                            stride=host.stride[0], upsample=kwargs.pop("upsample", False), lora=blk, **kwargs)
            return self._dropped(y, dblk, residual) if drop else y
        if kwargs:
            raise NotImplementedError(f"LoraHipContainer: unsupported call arguments {list(kwargs)}")
        if drop:
            return self._dropped(ops.linear(x, self._host, blk, None), dblk, residual)
        return ops.linear(x, self._host, blk, residual)

    @staticmethod
    def _dropped(y, blk, residual):
        """``dropout(layer(x, W_host + dW, bias))`` (lora_base_patch.py:74): the reference drops the WHOLE layer output, host path
        included, with torch's dropout (its Philox stream, its 1/(1-p) scaling) — reproduced with the same op; a residual that
        the native caller wanted fused into the GEMM epilogue is added afterwards (it is not part of the layer's output)."""
        y = blk.dropout(y)
        return y if residual is None else ops.add(y, residual)


class _Factors(nn.Module):
    """`layer` sub-module: holds W_down / W_up with the reference's names, shapes and init."""

    def __init__(self, in_features, out_features, rank, conv1x1=False):
        super().__init__()
        self.rank = rank
        tail = (1, 1) if conv1x1 else ()                 # a Conv2d host keeps the reference's 4-D factor shapes
        self.W_down = nn.Parameter(torch.empty(rank, in_features, *tail))
        self.W_up = nn.Parameter(torch.empty(out_features, rank, *tail))
        self.register_parameter("bias", None)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.W_down, a=math.sqrt(5))
        nn.init.zeros_(self.W_up)

    def get_weight(self):
        return torch.mm(self.W_up.flatten(1), self.W_down.flatten(1))

    def get_collapsed_param(self):
        return self.W_up.data.flatten(1) @ self.W_down.data.flatten(1), None


class _ConvFactors(nn.Module):
    """3x3 conv host (LoCon): W_down [r,Cin,3,3], W_up [Cout,r,1,1] — names / shapes / init of the reference's
    LoraLayer.Conv2dLayer (lora_layers_patch.py:64-100)."""

    def __init__(self, cin, cout, rank):
        super().__init__()
        self.rank = rank
        self.W_down = nn.Parameter(torch.empty(rank, cin, 3, 3))
        self.W_up = nn.Parameter(torch.empty(cout, rank, 1, 1))
        self.register_parameter("bias", None)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.W_down, a=math.sqrt(5))
        nn.init.zeros_(self.W_up)

    def get_weight(self):
        return torch.einsum("or,rikl->oikl", self.W_up[:, :, 0, 0], self.W_down)

    def get_collapsed_param(self):
        return torch.einsum("or,rikl->oikl", self.W_up.data[:, :, 0, 0], self.W_down.data), None


class LoraHipLayer(PatchPluginBlock):
    container_cls = LoraHipContainer
    wrapable_classes = (nn.Linear, nn.Conv2d)

    def __init__(self, lora_id, host, rank=1, dropout=0.1, alpha=1.0, bias=False, alpha_auto_scale=True, parent_block=None,
                 host_name=None, **kwargs):
        """Signature and defaults of the reference's ``LoraLayer`` (lora_layers_patch.py:21-23; ``dropout=0.1`` when the class is
        constructed directly).  The path every trainer config takes — make_hcpdiff -> ``wrap_model`` -> ``wrap_layer`` — passes
        ``wrap_layer``'s OWN default ``dropout=0.0`` (lora_base_patch.py:148-149) unless the cfg item names one, and so does this
        class: a cfg item without a ``dropout`` key trains without dropout in both code bases."""
        super().__init__(f"lora_block_{lora_id}", host, parent_block=parent_block, host_name=host_name)
        host = self.host()
        if not isinstance(host, (HipLinear, HipConv2d)):
            raise NotImplementedError(f"lora_hip: host {type(host).__name__} is not a native Linear / Conv2d layer")
        conv3 = isinstance(host, HipConv2d) and host.kernel_size == (3, 3)
        if bias:
            # unreachable in the reference as well: LinearLayer / Conv2dLayer.reset_parameters do `if self.bias:` on the [out]-element
            # Parameter (lora_layers_patch.py:41,85) -> "Boolean value of Tensor with more than one value is ambiguous"
            raise NotImplementedError("lora_hip: bias=True (the reference's own LoraLayer raises on it: lora_layers_patch.py:41)")
        out_f, in_f = host.weight.shape[0], host.weight.shape[1]
        # conv_in / conv_out (4 latent channels; their own NCHW <-> NHWC module classes): the side-path kernels want channel counts that
        # are multiples of 8, so these two small layers run the reference's own arithmetic instead — host convolution on the MERGED weight
        # W + sum alpha W_up W_down (lora_base_patch.py:20-35), factor gradients from the merged weight's gradient (ops.merged_lora_call)
        self.merged = bool(conv3 and (in_f % 8 or out_f % 8 or type(host) is not HipConv2d))
        if isinstance(rank, float):
            rank = max(round(out_f * rank), 1)            # fractional rank, lora_base_patch.py:105-106
        # rank <= 32: one MFMA k-step of rank slots, fused into the host GEMM.  Above (any rank: the reference's fractional ranks,
        # lora_base_patch.py:105-106, reach half the layer width): the side path runs as its own skinny GEMM (T = x W_down^T) and rides
        # into the host GEMM as a K-extension of ceil(rank/32)*32 columns (3x3 conv host: T = conv3x3(x, W_down) with ceil(rank/32)*32
        # output channels, added to the host convolution's output by one more GEMM).
        self.wide = rank > RANK_SLOT and not self.merged
        self.rank_pad = (rank + RANK_SLOT - 1) // RANK_SLOT * RANK_SLOT
        self.host_type = "conv" if conv3 else "linear"   # a 1x1 conv is a Linear on channels-last tokens
        self.bias = bias
        self.layer = (_ConvFactors(in_f, out_f, rank) if conv3 else
                      _Factors(in_f, out_f, rank, conv1x1=isinstance(host, HipConv2d))).to(host.weight.device)
        self.dropout = nn.Dropout(dropout)
        self.rank = rank
        self.register_buffer("alpha", torch.tensor(alpha / rank if alpha_auto_scale else alpha, device=host.weight.device))
        self.alpha_f = float(self.alpha)
        self._bucket = None       # set by LoraBucket.adopt
        self._pk = None

    # ---- reference API surface
    def get_weight(self):
        return self.layer.get_weight() * self.alpha

    def get_bias(self):
        return None

    def init_weights(self, svd_init=False):
        """svd_init (lora_base_patch.py:76-82): the factors start as the rank-r truncated SVD of the host weight, U S -> W_up,
        V^T -> W_down, both clamped at the 0.99 quantile of their joint value distribution (utils/utils.py:17-41).  The reference's
        own patch-LoRA cannot run this path (its feed_svd writes ``lora_up`` / ``lora_down``, attributes its layers do not have,
        lora_base_patch.py:108-110, and low_rank_approximate unpacks a flattened conv weight into four dims, utils.py:19-20): what
        is implemented is what those lines evidently mean.  A one-off host call (torch.linalg.svd), like checkpoint loading."""
        if not svd_init:
            self.layer.reset_parameters()
            return
        with torch.no_grad():
            w = self.host().weight.detach().float()
            w2 = w.flatten(1)                                            # conv: [out, in*kh*kw]
            U, S, Vh = torch.linalg.svd(w2, full_matrices=False)
            U = U[:, :self.rank] @ torch.diag(S[:self.rank])
            Vh = Vh[:self.rank]
            dist = torch.cat([U.flatten(), Vh.flatten()])
            if dist.numel() > (1 << 24):                                 # torch.quantile's input limit: a strided subsample
                dist = dist[::(dist.numel() >> 24) + 1]
            hi = torch.quantile(dist, 0.99)
            self.layer.W_up.copy_(U.clamp(-hi, hi).reshape(self.layer.W_up.shape))
            self.layer.W_down.copy_(Vh.clamp(-hi, hi).reshape(self.layer.W_down.shape))      # conv: [r, in*kh*kw] -> [r, in, kh, kw]

    def reparameterization_to_host(self, alpha=None, base_alpha=1.0):
        alpha = self.alpha if alpha is None else alpha
        host = self.host()
        w, _ = self.layer.get_collapsed_param()
        host.weight = nn.Parameter(host.weight.data * base_alpha + alpha * w.view_as(host.weight).to(host.weight))

    @classmethod
    def wrap_layer(cls, lora_id, layer, rank=1, dropout=0.0, alpha=1.0, svd_init=False, bias=False, mask=None, **kwargs):
        blk = cls(lora_id, layer, rank, dropout, alpha, bias=bias, **kwargs)
        blk.init_weights(svd_init)
        return blk

    @classmethod
    def wrap_model(cls, lora_id, model, **kwargs):
        return super().wrap_model(lora_id, model, exclude_classes=(LoraHipLayer,), **kwargs)

    # ---- native operand management
    def packed(self):
        if self._bucket is None:
            LoraBucket([self])                 # stand-alone layer: a private one-layer bucket
        return self._bucket.packed_for(self)

    def grad_views(self):
        if self._bucket is None:
            LoraBucket([self])
        return self._bucket.grad_views_for(self)

    def members(self):
        """[(block, first rank slot)] — one entry for a single block, several for MultiLora."""
        return [(self, 0)]


class DAPPHipContainer(LoraHipContainer):
    """Native twin of the reference's DAPPPatchContainer (lora_layers_patch.py:102-135, DreamArtist++): the batch is [negative half;
    positive half]; the first half runs the host with the summed 'n'-branch blocks, the second with the 'p'-branch blocks, and the halves
    are concatenated again.  Each half is the ordinary fused-LoRA call of LoraHipContainer on that branch's blocks (their rank slots side
    by side when a branch holds several).  Like the reference, both branches need at least one block (its `host_weight + None` raises
    otherwise, lora_base_patch.py:74), and the LAST plugin's dropout acts on both halves."""

    def forward(self, x, residual=None, **kwargs):
        names = {"p": tuple(n for n in self.plugin_names if self[n].branch == "p"),
                 "n": tuple(n for n in self.plugin_names if self[n].branch == "n")}
        if not names["p"] or not names["n"]:
            raise ValueError("dapp_hip: a host needs at least one 'p' and one 'n' branch block (the reference adds None to the host weight "
                             "otherwise, lora_layers_patch.py:131-133)")
        B = x.shape[0] // 2
        last = self[self.plugin_names[-1]]              # post_forward of the LAST plugin — its dropout — for BOTH halves (lora_layers_patch.py:132-133)
        half = lambda t, lo: None if t is None else (t[:B] if lo else t[B:])
        # per-sample call arguments of a conv host (ResnetBlock2D: rowbias = the time-embedding projection, x2 = a concatenated input) follow
        # their half of the batch; `upsample` is a flag
        kw = lambda lo: {k: (half(v, lo) if torch.is_tensor(v) else v) for k, v in kwargs.items()}
        y_n = self._run(names["n"], x[:B], half(residual, True), drop_block=last, **kw(True))
        y_p = self._run(names["p"], x[B:], half(residual, False), drop_block=last, **kw(False))
        return torch.cat([y_n, y_p], dim=0)


class DAPPHipLayer(LoraHipLayer):
    """``lora_layer_map['dapp_hip']``: LoraHipLayer with a ``branch`` ('p' / 'n'), reference DAPPLayer (lora_layers_patch.py:137-141).
    (The reference's own DAPPLayer derives from LoraBlock, which has no LinearLayer / Conv2dLayer classes, and cannot be constructed as
    shipped; the semantics implemented are those of its container.)"""
    container_cls = DAPPHipContainer

    def __init__(self, lora_id, host, rank=1, dropout=0.1, alpha=1.0, bias=False, alpha_auto_scale=True, branch="p", **kwargs):
        if branch not in ("p", "n"):
            raise ValueError(f"dapp_hip: branch must be 'p' or 'n', got {branch!r}")
        super().__init__(lora_id, host, rank, dropout, alpha, bias=bias, alpha_auto_scale=alpha_auto_scale, **kwargs)
        self.branch = branch

    @classmethod
    def wrap_layer(cls, lora_id, layer, rank=1, dropout=0.0, alpha=1.0, svd_init=False, bias=False, mask=None, branch="p", **kwargs):
        blk = cls(lora_id, layer, rank, dropout, alpha, bias=bias, branch=branch, **kwargs)
        blk.init_weights(svd_init)
        return blk


class _LoraOperands:
    pass


class MultiLora:
    """Several LoRA blocks on ONE host — the reference container sums ``get_weight()`` over all its plugins
    (LoraPatchContainer.forward, lora_base_patch.py:20-27).  Natively: one fused-LoRA GEMM (Linear host) or one skinny 3x3 convolution +
    K-extension (3x3 conv host) whose 32 rank slots hold the blocks' factors side by side (sum of ranks, each padded to 8, must fit 32)."""

    def __init__(self, blocks, names):
        self.blocks, self.names = list(blocks), names
        self.host_type = self.blocks[0].host_type
        if any(b.host_type != self.host_type for b in self.blocks):
            raise NotImplementedError("hcp_diffusion_amd: the LoRA blocks of one host must be of one kind")
        self.slot_off, sl = [], 0
        for b in self.blocks:
            self.slot_off.append(sl); sl += 8 * ((b.rank + 7) // 8)
        # more than 32 slots in total (two rank-32 LoRAs stacked, a wide block among them, ...): the blocks share the WIDE form instead —
        # one skinny side GEMM over all slots and a K-extension of ceil(slots / 32) * 32 columns (Linear hosts)
        self.wide = sl > RANK_SLOT
        self.rank_pad = (sl + RANK_SLOT - 1) // RANK_SLOT * RANK_SLOT
        if self.wide and self.host_type != "linear":
            raise NotImplementedError(f"hcp_diffusion_amd: LoRA blocks on one 3x3 conv host need {sl} > {RANK_SLOT} rank slots")
        buckets = {id(b._bucket) for b in self.blocks}
        if buckets == {id(None)}:
            LoraBucket(self.blocks)                        # stand-alone layers: one private bucket for the host's blocks
        elif len(buckets) != 1:
            raise NotImplementedError("hcp_diffusion_amd: LoRA blocks of one host must live in the same LoraBucket")
        self.bucket = self.blocks[0]._bucket
        self.ops = None
        self.bucket.add_multi(self)
        self.layer = self.blocks[0].layer                  # autograd anchor: any trainable factor pair

    def packed(self):
        return self.bucket.packed_group(self)

    def members(self):
        return list(zip(self.blocks, self.slot_off))


class FusedLoraGroup:
    """Several LoRA'd Linear layers that read the SAME input and are evaluated as ONE fused-LoRA GEMM
    (q/k/v of self-attention; k/v of cross-attention): outputs concatenated along N, the layers' rank slots packed
    side by side in one 32-wide slot group.  `blocks[i]` may be None for a host without LoRA."""

    def __init__(self, hosts, blocks, out_scale=None):
        """out_scale[i]: a constant folded into member i's output, y_i = c_i (x W_i^T + alpha_i T_i W_up_i^T) — in the packed host
        weight (one rounding of c W, as W itself) and in the member's LoRA alpha, forward and backward alike.  The attention modules use
        it to hand the kernels Q * d^-0.5 * log2(e) (no per-score multiply in front of the exponential)."""
        self.hosts, self.blocks = list(hosts), list(blocks)
        self.out_scale = [1.0] * len(self.hosts) if out_scale is None else [float(c) for c in out_scale]
        self.n_off, n = [], 0
        for h in self.hosts:
            self.n_off.append(n); n += h.weight.shape[0]
        self.n_total = n
        self.k = self.hosts[0].weight.shape[1]
        self.slot_off, sl = [], 0
        for b in self.blocks:
            self.slot_off.append(sl); sl += 8 * ((b.rank + 7) // 8) if b is not None else 0
        self.slots = sl
        self.has_lora = any(b is not None for b in self.blocks)
        self._pk = None
        self.ops = None         # _LoraOperands with the fused images (set by LoraBucket.add_group)

    def packed_host(self):
        """bf16 [Ntot, K] (forward B operand) and [K, Ntot] (dX B operand) of the concatenated frozen host weights."""
        key = tuple((h.weight._version, h.weight.data_ptr()) for h in self.hosts)
        if self._pk is None or self._pk[0] != key:
            w = torch.cat([h.weight.detach().reshape(h.weight.shape[0], -1) * c for h, c in zip(self.hosts, self.out_scale)], 0).to(BF16).contiguous()
            self._pk = (key, w, w.t().contiguous())
        return self._pk[1], self._pk[2]


class CtxBatch:
    """The K/V projection groups of ALL cross-attention layers of a model as ONE GEMM: they read the same tensor (the prompt states,
    wrapper.py:29 hands every layer the same encoder_hidden_states), so  [K_0|V_0|K_1|V_1|...] = [ctx | T_all] [W_all | BU]^T  with
    T_all = ctx AD_all^T (every layer's rank slots side by side, 32 columns per group) and BU block-diagonal (alpha W_up of group g in
    rows of group g, columns of group g; zeros elsewhere, written once).  32 small launches per forward -> 3.  Built by
    unet.NativeUNet2DConditionModel when every group is fusable and the prompt states need no gradient."""

    def __init__(self, groups):
        self.groups = list(groups)
        self.k = self.groups[0].k
        assert all(g.k == self.k for g in self.groups)
        self.n_off, n = [], 0
        for g in self.groups:
            self.n_off.append(n); n += g.n_total
        self.n_total = n
        buckets = [g.bucket for g in self.groups if g.has_lora]
        self.bucket = buckets[0] if buckets else None
        assert all(b is self.bucket for b in buckets), "one LoRA bucket per model"
        self.k2 = RANK_SLOT * len(self.groups) if self.bucket is not None else 0
        # split T (kernels.T_SPLIT): T_all travels as (hi | lo) — 2 * k2 extension columns against [BU | BU]
        self.split = bool(K.T_SPLIT and self.k2)
        self.kext = self.k2 * (2 if self.split else 1)
        dev = self.groups[0].hosts[0].weight.device
        self.ad_all = torch.zeros(max(self.k2, 1), self.k, dtype=BF16, device=dev) if self.k2 else None
        self.b_cat = torch.zeros(self.n_total, self.k + self.kext, dtype=BF16, device=dev)     # [W_all | BU (| BU)]: padding stays zero forever
        # backward: U_all = dY_all BUT^T, every layer's dY W_up in ONE deep-K GEMM (block-diagonal again)
        self.but_all = torch.zeros(max(self.k2, 1), self.n_total, dtype=BF16, device=dev) if self.k2 else None
        self._host_key = None
        self.refresh_hosts()
        if self.bucket is not None:
            self.bucket.add_ctx_batch(self)

    def refresh_hosts(self):
        """(Re)write the frozen host weights' columns of the joint operand when a host weight was replaced (load_state_dict)."""
        key = tuple((h.weight._version, h.weight.data_ptr()) for g in self.groups for h in g.hosts)
        if key != self._host_key:
            for g, off in zip(self.groups, self.n_off):
                self.b_cat[off:off + g.n_total, :self.k].copy_(g.packed_host()[0])
            self._host_key = key

    def operands(self):
        if self.bucket is not None and self.bucket._stale([b for g in self.groups for b in g.blocks]):
            self.bucket.pack()
        return self.ad_all, self.b_cat

    def all_have_lora(self):
        return all(b is not None for g in self.groups for b in g.blocks)


class LoraBucket:
    """All LoRA factors of a model in ONE flat fp32 parameter buffer + ONE flat gradient buffer.

    * W_down/W_up become views into `params`; their `.grad` are views into `grads` (so torch optimizers, the
      reference's clip_grad_norm_ and checkpointing still see ordinary parameters);
    * `pack()` refreshes the bf16 operand copies of every layer (and of every fused group) with one kernel launch
      (call once per step, after the optimizer update);
    * the data-parallel exchange is one all-reduce of `grads` (dist.py), the optimizer one fused launch (optim.py).
    """

    def __init__(self, blocks):
        self.blocks = list(blocks)
        dev = self.blocks[0].layer.W_down.device
        self.device = dev
        n = sum(b.layer.W_down.numel() + b.layer.W_up.numel() for b in self.blocks)
        self.params = torch.empty(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self._ops = {}
        self._gviews = {}          # the 2-D / channels-last tensors the weight-gradient kernels write
        self._pgrads = {}          # the same memory with the parameters' own shapes (their .grad)
        self._desc_bytes = bytearray()
        self._desc_count = 0
        self._images = []          # zero-initialised operand images (kept alive)
        self.groups = []
        self._conv_rows, self._conv_tiles, self._conv_pieces = [], 0, None
        for b in self.blocks:
            views = []
            for p in (b.layer.W_down, b.layer.W_up):
                seg, gseg = self.params[off:off + p.numel()], self.grads[off:off + p.numel()]
                if p.dim() == 4 and p.shape[2] == 3:       # conv W_down [r,Cin,3,3]: channels_last = the kernels' [r][ky][kx][Cin]
                    v = seg.view(p.shape[0], 3, 3, p.shape[1]).permute(0, 3, 1, 2)
                    g = gseg.view(p.shape[0], 3, 3, p.shape[1]).permute(0, 3, 1, 2)
                else:
                    v, g = seg.view_as(p), gseg.view_as(p)
                v.copy_(p.data)
                p.data = v
                p.grad = g
                views.append(g)
                off += p.numel()
            k = b.layer.W_down.shape[1]
            n_out = b.layer.W_up.shape[0]
            self._pgrads[id(b)] = tuple(views)
            if b.host_type == "conv":
                self._gviews[id(b)] = (views[0].permute(0, 2, 3, 1), views[1].view(n_out, -1))     # [r][3][3][Cin], [Cout, r]
                self._ops[id(b)] = None if b.merged else self._new_conv_images(b, k, n_out)
            elif b.wide:
                self._gviews[id(b)] = (views[0].view(views[0].shape[0], k), views[1].view(n_out, -1))
                self._ops[id(b)] = self._new_wide_images(b, k, n_out)
            else:
                self._gviews[id(b)] = (views[0].view(views[0].shape[0], k), views[1].view(n_out, -1))
                self._ops[id(b)] = self._new_images(k, n_out)
                self._add_desc(b, self._ops[id(b)], 0, 0, n_out)
            b._bucket = self
        assert K.lib().hcp_lora_pack_desc_bytes() == 80
        self._upload_descs()
        if self._conv_rows:
            import numpy as np
            from .fullft import PIECE_DTYPE
            arr = np.array(self._conv_rows, dtype=PIECE_DTYPE)
            self._conv_pieces = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        self._packed_version = None
        self.pack()

    def _new_images(self, k, n_total):
        o = _LoraOperands()
        img = torch.zeros(2 * RANK_SLOT * (k + n_total), dtype=BF16, device=self.device)   # padding stays zero forever
        self._images.append(img)
        a = RANK_SLOT * k; c = RANK_SLOT * n_total
        o.ad = img[0:a].view(RANK_SLOT, k); o.adt = img[a:2 * a].view(k, RANK_SLOT)
        o.bu = img[2 * a:2 * a + c].view(n_total, RANK_SLOT); o.but = img[2 * a + c:2 * a + 2 * c].view(RANK_SLOT, n_total)
        return o

    def _new_conv_images(self, b, cin, cout, slot0=0, into=None):
        """Operand images of a 3x3 conv LoRA block (zero padding written once) + the pack pieces that refresh them:
        ad [32][3][3][Cin] (T = conv3x3(x, W_down)), wdl [Cin][3][3][32] (its data gradient), bu [Cout][32] = alpha W_up,
        but [32][Cout].  slot0 / into: the block's factors go to rank slots [slot0, slot0 + r) of an EXISTING image set (several blocks
        on one host, MultiLora): rows slot0.. of ad / but, columns slot0.. of wdl / bu."""
        o = into
        rp = b.rank_pad if into is None else RANK_SLOT     # slots of the image set: 32, or the padded rank of a wide block (rank > 32)
        if o is None:
            o = _LoraOperands()
            img = torch.zeros(2 * rp * 9 * cin + 2 * rp * cout, dtype=BF16, device=self.device)
            self._images.append(img)
            a = rp * 9 * cin; c = rp * cout
            o.ad = img[0:a].view(rp, 3, 3, cin); o.wdl = img[a:2 * a].view(cin, 3, 3, rp)
            o.bu = img[2 * a:2 * a + c].view(cout, rp); o.but = img[2 * a + c:2 * a + 2 * c].view(rp, cout)
        r = b.layer.W_down.shape[0]
        wd = b.layer.W_down.permute(0, 2, 3, 1)            # physical [r][3][3][Cin] fp32
        assert wd.is_contiguous()
        tc = (cin + 63) // 64
        for tap in range(9):
            self._conv_rows.append((wd.data_ptr() + 4 * tap * cin, o.ad.data_ptr() + 2 * (slot0 * 9 * cin + tap * cin),
                                    o.wdl.data_ptr() + 2 * (tap * rp + slot0),
                                    r, cin, 9 * cin, 9 * cin, 9 * rp, self._conv_tiles, tc, 1.0))
            self._conv_tiles += ((r + 63) // 64) * tc
        tcu = (r + 63) // 64
        self._conv_rows.append((b.layer.W_up.data_ptr(), o.bu.data_ptr() + 2 * slot0, o.but.data_ptr() + 2 * slot0 * cout, cout, r, r, rp, cout,
                                self._conv_tiles, tcu, b.alpha_f))
        self._conv_tiles += ((cout + 63) // 64) * tcu
        return o

    def _new_wide_images(self, b, k, n_out, slot0=0, into=None, rp=None):
        """rank > 32 on a Linear host: ad [Rp][K] = W_down, wdt [K][Rp] = W_down^T, bu [N][Rp] = alpha W_up,
        but [Rp][N] = alpha W_up^T (Rp = rank padded to 32; padding stays zero), refreshed by two pack pieces.  slot0 / into / rp: the
        block's factors go to slots [slot0, slot0 + r) of an existing Rp-wide image set (several blocks on one host, MultiLora)."""
        o = into
        rp, r = (rp or b.rank_pad), b.rank
        if o is None:
            o = _LoraOperands()
            img = torch.zeros(2 * rp * (k + n_out), dtype=BF16, device=self.device)
            self._images.append(img)
            a = rp * k; c = rp * n_out
            o.ad = img[0:a].view(rp, k); o.wdt = img[a:2 * a].view(k, rp)
            o.bu = img[2 * a:2 * a + c].view(n_out, rp); o.but = img[2 * a + c:2 * a + 2 * c].view(rp, n_out)
        tc = (k + 63) // 64
        self._conv_rows.append((b.layer.W_down.data_ptr(), o.ad.data_ptr() + 2 * slot0 * k, o.wdt.data_ptr() + 2 * slot0, r, k, k, k, rp, self._conv_tiles, tc, 1.0))
        self._conv_tiles += ((r + 63) // 64) * tc
        tc = (r + 63) // 64
        self._conv_rows.append((b.layer.W_up.data_ptr(), o.bu.data_ptr() + 2 * slot0, o.but.data_ptr() + 2 * slot0 * n_out, n_out, r, r, rp, n_out, self._conv_tiles, tc, b.alpha_f))
        self._conv_tiles += ((n_out + 63) // 64) * tc
        return o

    def _add_desc(self, b, o, slot0, n0, n_total, alpha_mul=1.0):
        r, k = b.layer.W_down.shape[:2]
        n_out = b.layer.W_up.shape[0]
        self._desc_bytes += struct.pack("<6Q3if4i", b.layer.W_down.data_ptr(), b.layer.W_up.data_ptr(), o.ad.data_ptr(), o.adt.data_ptr(),
                                        o.bu.data_ptr(), o.but.data_ptr(), k, n_out, r, b.alpha_f * alpha_mul, slot0, n0, n_total, 0)
        self._desc_count += 1

    def _upload_descs(self):
        self.descs = torch.frombuffer(bytearray(self._desc_bytes or b"\0"), dtype=torch.uint8).to(self.device)

    def add_group(self, group):
        """Register a FusedLoraGroup: allocate its shared operand images and pack descriptors (one per member block)."""
        if group.slots > RANK_SLOT:
            raise ValueError("fused LoRA group needs more than 32 rank slots")
        group.ops = self._new_images(group.k, group.n_total)
        for b, n0, s0, c in zip(group.blocks, group.n_off, group.slot_off, group.out_scale):
            if b is not None:
                assert b._bucket is self
                self._add_desc(b, group.ops, s0, n0, group.n_total, alpha_mul=c)
        self._upload_descs()
        self.groups.append(group)
        self.pack()
        return group

    def add_ctx_batch(self, batch):
        """Pack descriptors that ALSO write every member block of the batch's groups into the batch's joint operands (CtxBatch)."""
        for gi, (g, goff) in enumerate(zip(batch.groups, batch.n_off)):
            for b, n0, s0, c in zip(g.blocks, g.n_off, g.slot_off, g.out_scale):
                if b is None:
                    continue
                assert b._bucket is self
                r, k = b.layer.W_down.shape[:2]
                n_out = b.layer.W_up.shape[0]
                self._desc_bytes += struct.pack("<6Q3if4i", b.layer.W_down.data_ptr(), b.layer.W_up.data_ptr(),
                                                batch.ad_all.data_ptr() + 2 * RANK_SLOT * gi * k, 0,
                                                batch.b_cat.data_ptr() + 2 * (batch.k + RANK_SLOT * gi),
                                                batch.but_all.data_ptr() + 2 * RANK_SLOT * gi * batch.n_total, k, n_out, r, b.alpha_f * c,
                                                s0, goff + n0, batch.n_total, batch.k + batch.kext)
                self._desc_count += 1
                if batch.split:                            # the second copy of alpha W_up, against the residual half of T_all
                    self._desc_bytes += struct.pack("<6Q3if4i", b.layer.W_down.data_ptr(), b.layer.W_up.data_ptr(), 0, 0,
                                                    batch.b_cat.data_ptr() + 2 * (batch.k + batch.k2 + RANK_SLOT * gi), 0, k, n_out, r,
                                                    b.alpha_f * c, s0, goff + n0, batch.n_total, batch.k + batch.kext)
                    self._desc_count += 1
        self._images += [batch.b_cat, batch.but_all]
        self._upload_descs()
        self.pack()

    def add_multi(self, multi):
        """Shared operand images for several blocks on one host (MultiLora): same output columns, adjacent rank slots."""
        b0 = multi.blocks[0]
        k, n_out = b0.layer.W_down.shape[1], b0.layer.W_up.shape[0]
        if multi.host_type == "conv":                      # one shared image set, every block's pack pieces aimed at its own slots
            multi.ops = None
            for b, s0 in zip(multi.blocks, multi.slot_off):
                assert b._bucket is self
                multi.ops = self._new_conv_images(b, k, n_out, slot0=s0, into=multi.ops)
            import numpy as np
            from .fullft import PIECE_DTYPE
            arr = np.array(self._conv_rows, dtype=PIECE_DTYPE)
            self._conv_pieces = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
            self.pack()
            return
        if multi.wide:                                     # shared wide images, every block's pack pieces aimed at its own slots
            multi.ops = None
            for b, s0 in zip(multi.blocks, multi.slot_off):
                assert b._bucket is self
                multi.ops = self._new_wide_images(b, k, n_out, slot0=s0, into=multi.ops, rp=multi.rank_pad)
            import numpy as np
            from .fullft import PIECE_DTYPE
            arr = np.array(self._conv_rows, dtype=PIECE_DTYPE)
            self._conv_pieces = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
            self.pack()
            return
        multi.ops = self._new_images(k, n_out)
        for b, s0 in zip(multi.blocks, multi.slot_off):
            assert b._bucket is self
            self._add_desc(b, multi.ops, s0, 0, n_out)
        self._upload_descs()
        self.pack()

    def pack(self):
        if self._desc_count:
            K.lora_pack(self.descs, self._desc_count)
        if self._conv_pieces is not None:                  # conv (LoCon) blocks: the grouped convert/transpose kernel
            K.pack_weights(self._conv_pieces, len(self._conv_rows), self._conv_tiles)
        self._packed_version = self.params._version
        ops.invalidate_merged_cache()                      # merged-weight hosts cache W + dW for gradient-free calls
        for b in self.blocks:                              # each factor's own counter too: an optimizer steps through the Parameters,
            b._pk_ver = (b.layer.W_down._version, b.layer.W_up._version)   # whose `.data` views do not share the bucket's version counter

    def _stale(self, blocks):
        """Has anything written the fp32 factors since the bf16 operands were derived?  The flat buffer's counter sees in-place
        writes to `params` (NativeTrainer's fused kernel is followed by an explicit pack()); the factors' own counters see a
        torch / reference-trainer optimizer stepping W_down / W_up in place (train_ac.py:491) — found by driving the reference's
        Trainer.train_one_step over these blocks: without this check step 2 ran on step-1 operands."""
        if self._packed_version != self.params._version:
            return True
        for b in blocks:
            if b is not None and getattr(b, "_pk_ver", None) != (b.layer.W_down._version, b.layer.W_up._version):
                return True
        return False

    def packed_for(self, blk):
        if self._stale((blk,)):
            self.pack()
        return self._ops[id(blk)]

    def packed_group(self, group):
        if self._stale(group.blocks):
            self.pack()
        return group.ops

    def grad_views_for(self, blk):
        gd, gu = self._gviews[id(blk)]
        if blk.layer.W_down.grad is None:      # zero_grad(set_to_none=True) (torch's default) dropped the views: that WAS the trainer's
            pg = self._pgrads[id(blk)]         # zero — clear the slices the kernels are about to accumulate into, then re-attach
            for g in pg:
                g.zero_()
            blk.layer.W_down.grad, blk.layer.W_up.grad = pg
        return gd, gu

    def zero_dropped(self):
        """Before a captured backward replays (graphed.py): clear and re-attach whatever `.grad` views the trainer dropped."""
        dropped = [b for b in self.blocks if b.layer.W_down.grad is None]
        if dropped and len(dropped) == len(self.blocks):
            self.grads.zero_()
        for b in dropped:
            pg = self._pgrads[id(b)]
            if len(dropped) != len(self.blocks):
                for g in pg:
                    g.zero_()
            b.layer.W_down.grad, b.layer.W_up.grad = pg

    @property
    def numel(self):
        return self.params.numel()


lora_layer_map = {"lora_hip": LoraHipLayer, "lora": LoraHipLayer, "dapp_hip": DAPPHipLayer, "dapp": DAPPHipLayer}

try:  # register with the reference's registry when it is importable (seam 2)
    from hcpdiff.models.lora_layers_patch import lora_layer_map as _ref_map   # pragma: no cover
    _ref_map.setdefault("lora_hip", LoraHipLayer)                              # pragma: no cover
    _ref_map.setdefault("dapp_hip", DAPPHipLayer)                              # pragma: no cover
except Exception:  # noqa: BLE001
    pass


def get_match_layers(patterns, named_modules):
    """`re:` / plain layer selectors of the reference configs (utils/cfg_net_tools.py:30-75, `re.match` anchored)."""
    import re
    out = []
    for pat in patterns:
        metas = pat.split(":")
        name = metas[-1]
        if "re" in metas[:-1]:
            rx = re.compile(name)
            out.extend(k for k in named_modules if rx.match(k) is not None)
        else:
            out.append(name)
    return sorted(set(out), key=out.index)


def make_lora(model, cfg_lora):
    """Restates make_hcpdiff's LoRA half (cfg_net_tools.py:108-123): wrap every matched layer, return
    ([param groups], PluginGroup, LoraBucket)."""
    named = dict(model.named_modules())
    groups, blocks = [], {}
    for lora_id, item in enumerate(cfg_lora):
        item = dict(item)
        layers = item.pop("layers")
        cls = lora_layer_map[item.pop("type", "lora_hip")]
        lr = item.pop("lr", None)             # None: the trainer's default lr
        params = []
        for layer_name in get_match_layers(layers, named):
            parent_name, _, host_name = layer_name.rpartition(".")
            made = cls.wrap_model(lora_id, named[layer_name], parent_block=named[parent_name], host_name=host_name, **item)
            for k, v in made.items():
                path = f"{layer_name}.{k}" if k else layer_name
                blocks[path] = v
                v.requires_grad_(True)
                v.train()
                params.extend(v.parameters())
        groups.append({"params": params, "lr": lr} if lr is not None else {"params": params})   # no key: a torch optimizer's own default applies
    bucket = LoraBucket(blocks.values()) if blocks else None
    return groups, PluginGroup(blocks), bucket
