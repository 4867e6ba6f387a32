"""NativeUNet2DConditionModel — seam 1 of the drop-in boundary: the object a reference config
injects as ``model.unet`` (hcpdiff/train_ac.py:220-222) and the trainer calls as
``unet(noisy_latents, timesteps, encoder_hidden_states, encoder_attention_mask=...).sample`` (models/wrapper.py:29).

Same module tree / parameter names as diffusers' ``UNet2DConditionModel`` (reference cfgs/unet_struct.txt), real
``nn.Linear`` / ``nn.Conv2d`` leaves called through ``__call__`` (forward hooks and LoRA containers fire), but every
op is a hand-written gfx950 kernel (ops.py -> C ABI).  Internally activations are bf16 channels-last
([B,H,W,C] == [B,HW,C] token-major, no permutes between conv and transformer blocks); NCHW exists only at
conv_in / conv_out.  GroupNorm+SiLU, bias, time-embedding add, residual adds, the skip concat and the nearest-2x
upsample are fused into the producing / consuming kernels where the module boundaries allow it.
"""
import os

import torch
from torch import nn

from . import kernels as K
from . import ops
from .layers import HipConv2d, HipConvIn, HipConvOut, HipGroupNorm, HipLayerNorm, HipLinear

BF16 = torch.bfloat16

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    num_attention_heads=8, cross_attention_dim=768, norm_num_groups=32, num_train_timesteps=1000,
    transformer_layers_per_block=1, use_linear_projection=False, addition_embed_type=None, addition_time_embed_dim=None,
    projection_class_embeddings_input_dim=None)

# SDXL-base (BASELINE.json configs[3]); the reference fixes only the call contract (models/wrapper.py:57-74) and the
# block-index map (tools/lora_convert.py:116-186) — the numbers are the public SDXL config.json.
SDXL_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    num_attention_heads=(5, 10, 20), cross_attention_dim=2048, norm_num_groups=32, num_train_timesteps=1000,
    transformer_layers_per_block=(1, 2, 10), use_linear_projection=True, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)


def _per_block(v, i):
    return v[i] if isinstance(v, (tuple, list)) else v


def _call_res(mod, x, residual):
    """Call a (possibly LoRA-wrapped / hooked) leaf, fusing the residual add when the callee supports it.
    residual = (hi, lo): a (hi | lo) residual stream (BasicTransformerBlock) — the result is a pair too.  A leaf that cannot take the
    pair (hooked, foreign, dropout, conv host, rank > 32) gets the hi image as an ordinary residual: that one add then rounds to bf16
    as the plain stream does, and the stream goes on without a lo image."""
    if isinstance(residual, tuple):
        leaf = _ff_out_leaf(mod)
        if leaf is not None:
            return ops.linear_stream(x, leaf[0], leaf[1], residual[0], residual[1])
        return _call_res(mod, x, residual[0]), None
    if getattr(mod, "supports_fused_residual", False):
        return mod(x, residual=residual)
    return ops.add(mod(x), residual)     # e.g. the reference's own LoraPatchContainer swallows extra kwargs


def _ckpt(module, fn, *args, **kwargs):
    """Gradient checkpointing at diffusers' granularity (one ResnetBlock2D / Transformer2DModel call per segment), always
    non-reentrant like the reference forces it (hcpdiff/train_ac.py:44-47).  A memory-for-recompute trade the reference
    defaults to on 24 GB GPUs (train_base.yaml:69); with 288 GB HBM it only costs an extra forward, so it stays off unless
    enable_gradient_checkpointing() is called."""
    if getattr(module, "gradient_checkpointing", False) and torch.is_grad_enabled():
        from torch.utils.checkpoint import checkpoint
        wg = ops.current_wgrad()                       # the recompute runs on the autograd thread: its nodes belong to the same step

        def run(*a, **k):
            with ops.wgrad_context(wg):
                return fn(*a, **k)
        return checkpoint(run, *args, use_reentrant=False, preserve_rng_state=False, **kwargs)
    return fn(*args, **kwargs)


def _tb(temb_act, resnet):
    """This resnet's slice of the batched time-embedding projection, if the UNet attached one to `temb_act`."""
    table = getattr(temb_act, "_hcp_tb", None)
    return table.get(id(resnet)) if table is not None else None


class Timesteps(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.num_channels = dim

    def forward(self, t):
        return K.timestep_embedding(t.contiguous(), self.num_channels)


class SiLU(nn.Module):
    def forward(self, x):
        return ops.silu(x)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = HipLinear(cin, dim)
        self.act = SiLU()
        self.linear_2 = HipLinear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps=1e-5):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, eps=eps)
        self.conv1 = HipConv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = HipLinear(temb_dim, cout)
        self.norm2 = HipGroupNorm(groups, cout, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = HipConv2d(cout, cout, 3, 1, 1)
        self.nonlinearity = SiLU()
        if cin != cout:
            self.conv_shortcut = HipConv2d(cin, cout, 1)
        self.gradient_checkpointing = False

    def forward(self, x, temb_act, skip=None, temb_bias=None):
        """x [B,H,W,C] (+ optional skip tensor = the up-path concat); temb_act = SiLU(temb); temb_bias = this block's
        slice of the batched time-embedding projection (fp32 [B,Cout]) when the UNet precomputed it."""
        if skip is not None:
            x = ops.concat_channels(x, skip)          # GroupNorm needs the joint tensor once; convs read it back
        # x continues as the residual (identity, or the input of the 1x1 shortcut conv): the fork hands the gradient arriving on
        # that path to norm1's backward kernel as its addend instead of leaving an aten::add to autograd
        h, x = self.norm1(x, silu=True, fork=True)
        if temb_bias is not None:
            tb = temb_bias
        elif (isinstance(self.time_emb_proj, HipLinear) and not temb_act.requires_grad
              and not self.time_emb_proj.weight.requires_grad):  # [B, Cout] fp32 row bias fused into conv1's epilogue
            tb = self.time_emb_proj(temb_act, out_f32=True)
        else:
            tb = self.time_emb_proj(temb_act).float()
        h = self.conv1(h, rowbias=tb)
        h = self.norm2(h, silu=True)
        sc = self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x
        return self.conv2(h, residual=sc)


LOG2E = 1.4426950408889634


def _fusable_linear(m):
    """(host, lora block or None) if `m` is a native bias-free Linear — bare or in a single-block native LoRA
    container — with no hooks attached (a hooked module must be called through __call__); else None."""
    from .lora import LoraHipContainer
    host, blk = m, None
    if isinstance(m, LoraHipContainer):
        if type(m) is not LoraHipContainer or len(m.plugin_names) != 1:      # (a subclass — DAPPHipContainer — splits the batch in its own forward)
            return None
        host, blk = m._host, m[m.plugin_names[0]]
        if m._forward_hooks or m._forward_pre_hooks:
            return None
    if type(host) is not HipLinear or host.bias is not None or host._forward_hooks or host._forward_pre_hooks:
        return None
    if host.weight.requires_grad:                      # trainable host: the fused (concatenated) operand would go stale
        return None
    if blk is not None and blk.wide:                   # rank > 32 runs as its own skinny GEMM, not in a shared slot group
        return None
    if blk is not None and blk.dropout.p > 0.0:        # dropout acts on this layer's whole output (lora_base_patch.py:74): own call
        return None
    return host, blk




def _ff_out_leaf(m):
    """(host, lora block | None) when `m` — FeedForward's output projection — is a native HipLinear, bare or in a native single-block
    LoRA container of rank <= 32 without active dropout, and nobody hooked it; else None (module-by-module path)."""
    from .lora import LoraHipContainer
    if isinstance(m, LoraHipContainer):
        if (type(m) is not LoraHipContainer or len(m.plugin_names) != 1 or m._forward_hooks or m._forward_pre_hooks or type(m._host) is not HipLinear
                or m._host._forward_hooks or m._host._forward_pre_hooks):
            return None
        blk = m[m.plugin_names[0]]
        if blk.wide or blk.host_type == "conv" or (blk.dropout.p > 0.0 and blk.training):
            return None
        return m._host, blk
    if type(m) is HipLinear and not (m._forward_hooks or m._forward_pre_hooks):
        return m, None
    return None


class CrossAttention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        self.to_q = HipLinear(dim, dim, bias=False)
        self.to_k = HipLinear(ctx_dim, dim, bias=False)
        self.to_v = HipLinear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([HipLinear(dim, dim), nn.Dropout(0.0)])
        self._groups = {}

    def _group(self, mods, scales=None):
        """FusedLoraGroup for projections that share an input, or None when any of them must stay a separate call.
        scales: per-member output constants (the q projection carries d^-0.5 * log2(e) for the attention kernels)."""
        from .lora import FusedLoraGroup
        key = tuple(id(m) for m in mods)
        hit = self._groups.get(key)
        if hit is not None:
            return hit[0]
        pairs = [_fusable_linear(m) for m in mods]
        group = None
        if all(p is not None for p in pairs):
            blocks = [p[1] for p in pairs]
            buckets = {id(b._bucket) for b in blocks if b is not None}
            ok = len(buckets) <= 1 and all(b is None or b._bucket is not None for b in blocks)
            if ok and sum(8 * ((b.rank + 7) // 8) for b in blocks if b is not None) <= 32:
                group = FusedLoraGroup([p[0] for p in pairs], blocks, scales)
                lb = next((b for b in blocks if b is not None), None)
                group.bucket = lb._bucket if lb is not None else None
                if lb is not None:
                    group.bucket.add_group(group)
        if len(self._groups) >= 6:                   # stale keys (layers wrapped / unwrapped since): drop the oldest
            self._groups.pop(next(iter(self._groups)))
        self._groups[key] = (group, mods)            # keeps the modules alive so the ids stay unique
        return group

    def forward(self, x, context=None, residual=None, key_bias=None):
        qc = self.head_dim ** -0.5 * LOG2E
        if context is None:
            # q|k|v in one fused-LoRA GEMM, attention reads the slices in place; the q third comes out as q * d^-0.5 * log2(e)
            # (folded into the packed q weights and that block's LoRA alpha): the kernels exponentiate the accumulator as it is
            g = self._group((self.to_q, self.to_k, self.to_v), (qc, 1.0, 1.0))
            gq = gkv = None
            if g is None:                              # the three blocks' rank slots exceed one 32-wide slot group (rank 16: SDXL's configs[3]):
                gkv = self._group((self.to_k, self.to_v))                  # q alone + k|v together, the cross-attention form with context = x
                gq = self._group((self.to_q,), (qc,)) if gkv is not None else None
            if g is not None:
                o = ops.attention_packed(ops.linear_group(x, g), None, self.heads, q_prescaled=True)
            elif gq is not None:                       # 2 GEMMs + 1 gradient add instead of 3 + 2, and the pre-scaled-Q attention kernels
                o = ops.attention_packed(ops.linear_group(x, gq), ops.linear_group(x, gkv), self.heads, None, q_prescaled=True)
            else:
                o = ops.attention(self.to_q(x), self.to_k(x), self.to_v(x), self.heads)
        else:
            g = self._group((self.to_k, self.to_v))
            gq = self._group((self.to_q,), (qc,)) if (g is not None and key_bias is None) else None
            pre = getattr(context, "_hcp_kv", None)    # every layer's k|v from ONE GEMM over the shared prompt states (unet._batched_ctx_kv)
            kv = pre.get(id(self)) if (pre is not None and g is not None) else None
            if g is not None and kv is None:
                kv = ops.linear_group(context, g)
            if g is not None and gq is not None:       # unmasked cross-attention: pre-scaled q from a one-member group
                o = ops.attention_packed(ops.linear_group(x, gq), kv, self.heads, None, q_prescaled=True)
            elif g is not None:
                o = ops.attention_packed(self.to_q(x), kv, self.heads, key_bias)
            else:
                o = ops.attention(self.to_q(x), self.to_k(context), self.to_v(context), self.heads, key_bias)
        return _call_res(self.to_out[0], o, residual) if residual is not None else self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = HipLinear(dim, inner * 2)

    def forward(self, x):
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    geglu_in_epilogue = False      # see forward()

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), HipLinear(dim * 4, dim)])

    def forward(self, x, residual=None):
        leaf = _ff_out_leaf(self.net[2]) if not (self.net[0]._forward_hooks or self.net[0]._forward_pre_hooks) and self.net[1].p == 0.0 else None
        if leaf is not None and type(self.net[0]) is GEGLU:
            # GEGLU + output projection as one autograd node: the GEGLU backward rides in the epilogue of the projection's
            # input-gradient GEMM (ops._GegluLinearFn).  A hooked / foreign / dropout leaf keeps the module-by-module path below.
            pleaf = _ff_out_leaf(self.net[0].proj) if self.geglu_in_epilogue else None
            if pleaf is not None:
                # OPT-IN (FeedForward.geglu_in_epilogue / unet.set_geglu_epilogue): the GEGLU product out of the PROJECTION's epilogue,
                # formed from the fp32 (h | g) — one rounding instead of two (what the reference's fp32-returning LoRA layer gives GEGLU),
                # no stand-alone pass, 16 launches fewer per SD1.5 step.  Off by default: same-box A/B +0.3 % (SD1.5) / +1.7 % (SDXL) —
                # the projection's epilogue is where that kernel already spends its time (K = C, N = 8 C) — and the SDXL parity
                # ratios do not move (profiles/r6_ab_geglu_epilogue.txt)
                hg, act = ops.linear_geglu(x, pleaf[0], pleaf[1])
                return ops.geglu_linear(hg, leaf[0], leaf[1], residual, gact=act)
            return ops.geglu_linear(self.net[0].proj(x), leaf[0], leaf[1], residual)
        h = self.net[0](x)
        return _call_res(self.net[2], h, residual) if residual is not None else self.net[2](h)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.attn1 = CrossAttention(dim, dim, heads)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, ctx_dim, heads)
        self.norm1 = HipLayerNorm(dim)
        self.norm2 = HipLayerNorm(dim)
        self.norm3 = HipLayerNorm(dim)

    def forward(self, x, context):
        """x: [B, N, C] bf16, or the (hi, lo) pair of a (hi | lo) residual stream (Transformer2DModel.hi_lo_stream) — the block then
        returns a pair: the three residual adds and the three norms see x = hi + lo, 16 mantissa bits instead of 8."""
        context, key_bias = context if isinstance(context, tuple) else (context, None)   # (states, additive key mask)
        if isinstance(x, tuple):
            h, hi, lo = ops.layernorm_fork_stream(x[0], x[1], self.norm1)
            x = self.attn1(h, residual=(hi, lo))
            h, hi, lo = ops.layernorm_fork_stream(x[0], x[1], self.norm2)
            x = self.attn2(h, context, residual=(hi, lo), key_bias=key_bias)
            h, hi, lo = ops.layernorm_fork_stream(x[0], x[1], self.norm3)
            return self.ff(h, residual=(hi, lo))
        h, x = self.norm1(x, fork=True)                        # (LN(x), x): the fork fuses the residual-gradient add
        x = self.attn1(h, residual=x)
        h, x = self.norm2(x, fork=True)
        x = self.attn2(h, context, residual=x, key_bias=key_bias)
        h, x = self.norm3(x, fork=True)
        return self.ff(h, residual=x)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, ctx_dim, heads, groups, depth=1, linear_proj=False):
        super().__init__()
        self.norm = HipGroupNorm(groups, dim, eps=1e-6)
        # use_linear_projection (SDXL): an nn.Linear on tokens — the same GEMM on channels-last memory, but the leaf class
        # (and so the weight shape [C,C] vs [C,C,1,1] and what a LoRA regex over nn.Linear matches) follows diffusers.
        self.proj_in = HipLinear(dim, dim) if linear_proj else HipConv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, ctx_dim, heads) for _ in range(depth)])
        self.proj_out = HipLinear(dim, dim) if linear_proj else HipConv2d(dim, dim, 1)
        self.gradient_checkpointing = False
        # (hi | lo) residual stream through the transformer blocks: "auto" = on for stacks of two or more blocks (SDXL: 2 / 10 per level),
        # where the bf16 rounding of every residual add compounds; True / False force it.  The reference's LoRA layers keep this stream
        # in fp32 under autocast (their mm + fp32 bias promotes, lora_layers_patch.py:50-57); DESIGN section 4.
        self.hi_lo_stream = "auto"

    def _use_stream(self):
        if self.hi_lo_stream == "auto":
            return len(self.transformer_blocks) >= 2
        return bool(self.hi_lo_stream)

    def forward(self, x, context):
        B, H, W, C = x.shape
        h, x = self.norm(x, silu=False, fork=True)
        h = self.proj_in(h.view(B, H * W, C)) if isinstance(self.proj_in, nn.Linear) else self.proj_in(h).view(B, H * W, C)
        if self._use_stream():
            h = (h, None)                                      # the stream starts here: proj_in's output is the first hi image
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if isinstance(h, tuple):
            h = h[0]                                           # proj_out reads bf16(x), as autocast casts the reference's fp32 stream
        if isinstance(self.proj_out, nn.Linear):
            return _call_res(self.proj_out, h, x.view(B, H * W, C)).view(B, H, W, C)
        return _call_res(self.proj_out, h.view(B, H, W, C), x)


def _transformer(c, cfg, level):
    return Transformer2DModel(c, cfg["cross_attention_dim"], _per_block(cfg["num_attention_heads"], level), cfg["norm_num_groups"],
                              _per_block(cfg["transformer_layers_per_block"], level), cfg["use_linear_projection"])


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(x, upsample=True)       # nearest-2x folded into the conv's gather


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, n_layers, cfg, has_attn, add_down, level=0):
        super().__init__()
        g = cfg["norm_num_groups"]
        if has_attn:
            self.attentions = nn.ModuleList([_transformer(cout, cfg, level) for _ in range(n_layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, g) for i in range(n_layers)])
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_attn = has_attn
        self.gradient_checkpointing = False

    def forward(self, h, temb_act, context):
        skips = ()
        for i, res in enumerate(self.resnets):
            h = _ckpt(self, res, h, temb_act, temb_bias=_tb(temb_act, res))
            if self.has_attn:
                h = _ckpt(self, self.attentions[i], h, context)
            skips += (h,)
        if hasattr(self, "downsamplers"):
            h = self.downsamplers[0](h)
            skips += (h,)
        return h, skips


class CrossAttnDownBlock2D(_DownBlock):
    pass


class DownBlock2D(_DownBlock):
    pass


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, temb_dim, cfg):
        super().__init__()
        g = cfg["norm_num_groups"]
        self.attentions = nn.ModuleList([_transformer(c, cfg, len(cfg["block_out_channels"]) - 1)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, g), ResnetBlock2D(c, c, temb_dim, g)])
        self.gradient_checkpointing = False

    def forward(self, h, temb_act, context):
        h = self.resnets[0](h, temb_act, temb_bias=_tb(temb_act, self.resnets[0]))
        h = _ckpt(self, self.attentions[0], h, context)
        return _ckpt(self, self.resnets[1], h, temb_act, temb_bias=_tb(temb_act, self.resnets[1]))


class _UpBlock(nn.Module):
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
        self.gradient_checkpointing = False

    def forward(self, h, skips, temb_act, context):
        for i, res in enumerate(self.resnets):
            h = _ckpt(self, res, h, temb_act, skip=skips[-1], temb_bias=_tb(temb_act, res))
            skips = skips[:-1]
            if self.has_attn:
                h = _ckpt(self, self.attentions[i], h, context)
        if hasattr(self, "upsamplers"):
            h = self.upsamplers[0](h)
        return h


class UpBlock2D(_UpBlock):
    pass


class CrossAttnUpBlock2D(_UpBlock):
    pass


class UNet2DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


class _Config(dict):
    __getattr__ = dict.__getitem__


class NativeUNet2DConditionModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        cfg = dict(SD15_CONFIG, **cfg)
        self.config = _Config(cfg)
        boc = cfg["block_out_channels"]
        temb_dim = boc[0] * 4
        self.conv_in = HipConvIn(cfg["in_channels"], boc[0], 3, 1, 1)
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim)
        n = cfg["layers_per_block"]
        downs, out_c = [], boc[0]
        for i, t in enumerate(cfg["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            cls = CrossAttnDownBlock2D if t.startswith("CrossAttn") else DownBlock2D
            downs.append(cls(in_c, out_c, temb_dim, n, cfg, t.startswith("CrossAttn"), i != len(boc) - 1, level=i))
        self.down_blocks = nn.ModuleList(downs)
        ups, rev = [], list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            cls = CrossAttnUpBlock2D if t.startswith("CrossAttn") else UpBlock2D
            ups.append(cls(in_c, out_c, prev, temb_dim, n + 1, cfg, t.startswith("CrossAttn"), i != len(boc) - 1,
                           level=len(boc) - 1 - i))
        self.up_blocks = nn.ModuleList(ups)
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb_dim, cfg)
        self.conv_norm_out = HipGroupNorm(cfg["norm_num_groups"], boc[0], eps=1e-5)
        self.conv_act = SiLU()
        self.conv_out = HipConvOut(boc[0], cfg["out_channels"], 3, 1, 1)
        self.class_embedding = None
        if cfg["addition_embed_type"] == "text_time":
            self.add_time_proj = Timesteps(cfg["addition_time_embed_dim"])
            self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], temb_dim)
        elif cfg["addition_embed_type"] is not None:
            raise NotImplementedError(f"hcp_diffusion_amd: addition_embed_type={cfg['addition_embed_type']!r}")

    # ---- reference-facing conveniences (train_ac.py:257-278, wrapper.py:39-49)
    @classmethod
    def from_config(cls, config=None, **kw):
        return cls(**dict(config or {}, **kw))

    @classmethod
    def from_pretrained(cls, path=None, subfolder=None, pretrained_model_name_or_path=None, hip_graph=False, hip_graph_max_signatures=None, **kw):
        """Load diffusers-format weights (config.json + *.safetensors) — names are identical by construction.
        (``pretrained_model_name_or_path``: diffusers' own keyword, as the YAML overlays of cfgs/train/mi355x pass it;
        ``hip_graph: True`` = enable_hip_graph(): the module replays captured graphs under the reference's eager trainer loop;
        ``hip_graph_max_signatures: N`` = how many aspect-ratio bucket shapes keep their captured pair, default graphed.MAX_SIGNATURES.)"""
        path = path if path is not None else pretrained_model_name_or_path
        import json
        import os
        root = os.path.join(path, subfolder) if subfolder else path
        cfg = json.load(open(os.path.join(root, "config.json")))
        heads = cfg.get("num_attention_heads") or cfg.get("attention_head_dim", 8)    # diffusers' historical mis-naming
        tl = cfg.get("transformer_layers_per_block", 1)
        model = cls(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                    block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                    down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]),
                    num_attention_heads=heads if isinstance(heads, int) else tuple(heads), cross_attention_dim=cfg["cross_attention_dim"],
                    norm_num_groups=cfg["norm_num_groups"], transformer_layers_per_block=tl if isinstance(tl, int) else tuple(tl),
                    use_linear_projection=bool(cfg.get("use_linear_projection", False)),
                    addition_embed_type=cfg.get("addition_embed_type"), addition_time_embed_dim=cfg.get("addition_time_embed_dim"),
                    projection_class_embeddings_input_dim=cfg.get("projection_class_embeddings_input_dim"))
        from safetensors.torch import load_file
        model.load_state_dict(load_file(os.path.join(root, "diffusion_pytorch_model.safetensors")))
        if hip_graph:
            model.enable_hip_graph(max_signatures=hip_graph_max_signatures)
        return model

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None                                   # attention is always the fused flash kernel

    def enable_gradient_checkpointing(self):
        """diffusers API used by the reference wrapper (models/wrapper.py:39-49): every block with a
        ``gradient_checkpointing`` attribute recomputes its ResnetBlock2D / Transformer2DModel segments in backward."""
        for m in self.modules():
            if hasattr(m, "gradient_checkpointing"):
                m.gradient_checkpointing = True
        self._hcp_capturable = None

    def set_residual_stream(self, mode="auto"):
        """(hi | lo) residual stream of the transformer blocks (Transformer2DModel.hi_lo_stream): "auto" (stacks of >= 2 blocks), True, False."""
        for m in self.modules():
            if isinstance(m, Transformer2DModel):
                m.hi_lo_stream = mode
        self._hcp_capturable = None

    def set_geglu_epilogue(self, on=True):
        """Opt-in: form the GEGLU product in the FF projection's GEMM epilogue (FeedForward.forward)."""
        for m in self.modules():
            if isinstance(m, FeedForward):
                m.geglu_in_epilogue = bool(on)
        self._hcp_capturable = None

    def disable_gradient_checkpointing(self):
        for m in self.modules():
            if hasattr(m, "gradient_checkpointing"):
                m.gradient_checkpointing = False
        self._hcp_capturable = None

    def _batched_time_proj(self, temb_act):
        """All ResnetBlock2D.time_emb_proj layers read the same SiLU(temb): evaluate them as ONE GEMM against the
        concatenated (frozen) weights and hand each block its fp32 column slice (22 tiny launches -> 1)."""
        res = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        ok = (not temb_act.requires_grad) and all(
            type(r.time_emb_proj) is HipLinear and not r.time_emb_proj.weight.requires_grad and not r.time_emb_proj._forward_hooks
            and not r.time_emb_proj._forward_pre_hooks for r in res)
        if not ok:
            return
        key = tuple((r.time_emb_proj.weight._version, r.time_emb_proj.weight.data_ptr()) for r in res)
        if getattr(self, "_tb_cache", None) is None or self._tb_cache[0] != key:
            w = torch.cat([r.time_emb_proj.weight.detach() for r in res], 0).to(BF16).contiguous()
            b = torch.cat([r.time_emb_proj.bias.detach().float() for r in res], 0).contiguous()
            offs, o = [], 0
            for r in res:
                offs.append(o); o += r.time_emb_proj.weight.shape[0]
            self._tb_cache = (key, w, b, offs)
        _, w, b, offs = self._tb_cache
        allp = K.gemm(temb_act.reshape(-1, temb_act.shape[-1]), w, bias=b, out_f32=True)        # [B, sum Cout] fp32
        temb_act._hcp_tb = {id(r): allp[:, o:o + r.time_emb_proj.weight.shape[0]] for r, o in zip(res, offs)}

    def _batched_ctx_kv(self, ctx):
        """Every cross-attention layer projects the SAME prompt states to its keys and values: when all of those projection pairs are
        fusable (frozen bias-free hosts, bare or with one native LoRA block of rank <= 16 each, no dropout, no hooks) and the states need
        no gradient, evaluate them as one GEMM up front (lora.CtxBatch: 32 launches -> 3 for SD1.5) and hand each layer its column slice."""
        if not torch.is_tensor(ctx) or ctx.requires_grad:
            return
        pairs = [(m.attn2, m.attn2._group((m.attn2.to_k, m.attn2.to_v))) for m in self.modules() if isinstance(m, BasicTransformerBlock)]
        pairs = [(a, g) for a, g in pairs if g is not None and g.k == ctx.shape[-1]]      # (trainable / hooked layers keep their own call)
        if len(pairs) < 2:
            return
        xs, groups = [a for a, _ in pairs], [g for _, g in pairs]
        if len({id(g.bucket) for g in groups if g.has_lora}) > 1:
            return
        key = tuple(id(g) for g in groups) + (str(ctx.device),)
        hit = getattr(self, "_ctx_batch", None)
        if hit is None or hit[0] != key:
            if ctx.is_cuda and torch.cuda.is_current_stream_capturing():
                return                                  # (built by the warm-up; never allocate / pack under capture)
            from .lora import CtxBatch
            hit = (key, CtxBatch(groups), groups)       # keeps the groups alive so the ids stay unique
            self._ctx_batch = hit
        batch = hit[1]
        if not (ctx.is_cuda and torch.cuda.is_current_stream_capturing()):
            batch.refresh_hosts()
        kvs = ops.ctx_kv(ctx, batch)
        ctx._hcp_kv = {id(a): kv for a, kv in zip(xs, kvs)}

    def enable_hip_graph(self, on=True, _recorded_on_cpu=False, max_signatures=None):
        """Replay the forward and the backward of `unet(...)` as captured hipGraphs when an ordinary trainer calls the module in grad
        mode (LoRA and / or host parameters training, alone or under torch DDP; see graphed.py).  Call again (or `reset_hip_graph()`)
        after adding / removing LoRA layers or changing what trains.  `_recorded_on_cpu`: tests only — the same wiring with recorded
        callables in place of graphs on the interpreter backend."""
        self._hip_graph, self._hip_graphs, self._hip_graph_cpu = bool(on), {}, bool(_recorded_on_cpu)
        self._hcp_capturable = None
        from . import graphed
        graphed.set_max_signatures(self, max_signatures)

    def reset_hip_graph(self):
        self._hip_graphs = {}
        self._hcp_capturable = None

    def forward(self, sample, timestep, encoder_hidden_states, encoder_attention_mask=None, added_cond_kwargs=None,
                cross_attention_kwargs=None, **kwargs):
        if (getattr(self, "_hip_graph", False) and torch.is_grad_enabled() and torch.is_tensor(timestep)
                and ((sample.is_cuda and not torch.cuda.is_current_stream_capturing()) or getattr(self, "_hip_graph_cpu", False))):
            from . import graphed
            ok, ckpt = graphed.capturable_cached(self)
            if ok:
                added = added_cond_kwargs or {}
                ins = [sample, timestep, encoder_hidden_states, encoder_attention_mask, added.get("text_embeds"), added.get("time_ids")]
                key = tuple(None if t is None else (tuple(t.shape), t.dtype, bool(t.requires_grad)) for t in ins) + (ckpt,)

                def fwd(s_, t_, e_, m_, te_, ti_):
                    ak = dict(text_embeds=te_, time_ids=ti_) if te_ is not None else None
                    return self._forward_impl(s_, t_, e_, m_, ak).sample
                return UNet2DConditionOutput(graphed.call(self, ins, fwd, self._hip_graphs, key))
        return self._forward_impl(sample, timestep, encoder_hidden_states, encoder_attention_mask, added_cond_kwargs)

    def _forward_impl(self, sample, timestep, encoder_hidden_states, encoder_attention_mask=None, added_cond_kwargs=None):
        text_time = self.config["addition_embed_type"] == "text_time"
        if bool(added_cond_kwargs) != text_time:
            raise ValueError("hcp_diffusion_amd: added_cond_kwargs={text_embeds,time_ids} is required by (and only by) a UNet with "
                             "addition_embed_type='text_time' (SDXL; reference models/wrapper.py:66-73)")
        if sample.dtype == torch.float16 or encoder_hidden_states.dtype == torch.float16 or \
                (sample.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16):
            # the reference's default mixed_precision (cfgs/train/train_base.yaml:2, train_ac.py:116-123): no fp16 operand packing and no
            # loss-scale handling exist on this path — refuse instead of computing in another precision than the config names
            raise NotImplementedError("hcp_diffusion_amd: mixed_precision 'fp16' is not supported by the native MI355X path (bf16 compute, fp32 "
                                      "accumulation / masters): set mixed_precision: 'bf16' (INTEGRATION.md, Precision)")
        B = sample.shape[0]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64, device=sample.device)
        timestep = timestep.to(torch.int64).reshape(-1).expand(B).contiguous()
        temb = self.time_embedding(self.time_proj(timestep))
        if text_time:                                  # SDXL micro-conditioning: sinusoid(6 scalars) | pooled text -> MLP -> +temb
            text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            te = self.add_time_proj(time_ids.reshape(-1).float()).view(B, -1)
            add = ops.concat_channels(text_embeds.to(BF16).contiguous(), te)
            temb = ops.add(temb, self.add_embedding(add))
        temb_act = ops.silu(temb)                      # every ResnetBlock applies SiLU(temb): do it once
        self._batched_time_proj(temb_act)
        ctx = encoder_hidden_states
        if ctx.dtype != BF16:
            ctx = ctx.to(BF16)
        ctx = ctx.contiguous()
        if ctx is encoder_hidden_states and not ctx.requires_grad:
            ctx = ctx.detach()                         # (a private alias: _batched_ctx_kv hangs this call's k|v tensors on it)
        self._batched_ctx_kv(ctx)
        if encoder_attention_mask is not None:         # [B,L] 1 = attend: diffusers turns it into (1 - mask) * -10000, added
            m = encoder_attention_mask                 # to the cross-attention scores of every head / query [ext]
            if m.shape != ctx.shape[:2]:
                raise ValueError(f"encoder_attention_mask {tuple(m.shape)} does not match encoder_hidden_states {tuple(ctx.shape[:2])}")
            ctx = (ctx, ((1.0 - m.to(torch.float32)) * -10000.0).contiguous())
        # NativeTrainer(overlap_exchange=True): callbacks that fire when backward has finished every gradient of up_blocks + the
        # output head ('after_up': the gradient of the mid block's output is complete) and of the mid block ('after_mid')
        marks = getattr(self, "_bwd_marks", None)
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, s = blk(h, temb_act, ctx)
            skips += s
        if marks and h.requires_grad:
            h.register_hook(marks["after_mid"])
        h = self.mid_block(h, temb_act, ctx)
        if marks and h.requires_grad:
            h.register_hook(marks["after_up"])
        for blk in self.up_blocks:
            k = len(blk.resnets)
            h = blk(h, skips[-k:], temb_act, ctx)
            skips = skips[:-k]
        h = self.conv_norm_out(h, silu=True)
        return UNet2DConditionOutput(self.conv_out(h))
