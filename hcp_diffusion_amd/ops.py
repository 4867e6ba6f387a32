"""torch.autograd.Function wrappers: each forward/backward is one or a few C-ABI kernel launches (kernels.py).

Activations are bf16, token-major / NHWC.  Frozen-weight layers produce input gradients only; LoRA factors get
their gradients accumulated by the wgrad kernel straight into the flat fp32 bucket (their ``.grad`` views).
Host parameters with ``requires_grad`` (full fine-tuning, reference DreamBooth.yaml:6-10) get theirs from the TN-GEMM /
column-sum / norm-affine kernels, accumulated in place into ``param.grad`` (fp32; 3x3 conv weights in channels_last
storage = the kernels' [Cout][ky][kx][Cin]) — autograd sees ``None`` for them.
"""
import os
import threading
from contextlib import contextmanager

import torch

from . import kernels as K

BF16 = torch.bfloat16

# ---- how the LoRA weight gradients of a backward pass are launched ------------------------------------------------
# dW_down / dW_up are leaves of the backward graph (only the optimizer reads them), while the dX chain is a long sequence of
# small, latency-bound kernels.  A WgradContext says what a layer's backward does with its (U, x, T, dY):
#   immediate   one launch per layer (the default: any trainer, any thread, nothing to flush)
#   grouped     collect every layer's operands; flush() computes all pairs in ONE launch at the end of the backward pass
#   side        each layer's launch goes to a second HIP stream ordered behind the dX kernel by an event; join() at the end
#               (under hipGraph capture: a parallel branch; measured slower on ROCm 7.2, kept as an option)
# The context is an OBJECT owned by whoever drives the step (NativeTrainer, a graphed-module entry), made current around the
# FORWARD (`with wgrad_context(c): pred = unet(...)`); every autograd node records the context it was built under and its backward —
# on the autograd engine's thread — talks to that object only.  No process-global flag is read or flipped during backward: two
# trainers (a UNet and a second model) in one process cannot see each other's settings.


class WgradContext:
    def __init__(self, grouped=False, side_stream=False):
        self.grouped, self.side = bool(grouped), bool(side_stream)
        self.items, self.keep = [], []            # grouped: pending operands / what the last grouped launch keeps alive (hipGraph replays)
        self.stream, self.side_keep = None, []

    def add(self, U, x2, gd, T, dy2, gu, rank, alpha, slot0=0, ulo=0, tlo=0):
        """ulo / tlo: column offset of the residual half of a split U / T (K.t_lo: the fused-LoRA GEMMs' [M, 64] = (hi | lo)), 0 = none."""
        if self.grouped:
            self.items.append((U, x2, gd, T, dy2, gu, rank, alpha, slot0, ulo, tlo))
            return
        if slot0 != 0 or not dy2.is_contiguous() or (ulo != 0) != (U.shape[1] == 64) or (tlo != 0) != (T.shape[1] == 64):
            # member of a fused group (or mixed formats): the grouped entry point handles slots / strides
            self.keep = [K.lora_wgrad_grouped([(U, x2, gd, T, dy2, gu, rank, alpha, slot0, ulo, tlo)])]
            return
        if not (self.side and x2.is_cuda):
            K.lora_wgrad_pair(U, x2, gd, T, dy2, gu, rank, alpha)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=x2.device)
        ev = torch.cuda.Event()
        ev.record()                                   # after the dX kernel (U is complete)
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            K.lora_wgrad_pair(U, x2, gd, T, dy2, gu, rank, alpha)
        self.side_keep.append((U, x2, T, dy2))        # the allocator must not recycle these before the side kernel ran

    def flush(self):
        """grouped: all collected layers' weight gradients as one launch."""
        if self.items:
            keep = K.lora_wgrad_grouped(self.items)
            self.keep = [keep, self.items]            # descriptor table + operand tensors stay alive (hipGraph replays)
            self.items = []

    def join(self):
        """side: make the current stream wait for every launch on the side stream; release the tensors kept alive for it."""
        if self.stream is not None and self.side_keep:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.side_keep.clear()


_IMMEDIATE = WgradContext()
_tls = threading.local()


def current_wgrad():
    return getattr(_tls, "ctx", None) or _IMMEDIATE


@contextmanager
def wgrad_context(ctx):
    prev = getattr(_tls, "ctx", None)
    _tls.ctx = ctx
    try:
        yield ctx
    finally:
        _tls.ctx = prev


def _tr(p):
    """The parameter itself when it is trainable (it then becomes an input of the autograd node so that backward runs
    even if no activation input requires grad), else None."""
    return p if (p is not None and p.requires_grad and torch.is_grad_enabled()) else None


def grad_buffer(p, conv3x3=False):
    """fp32 accumulation target for parameter `p`: its ``.grad`` (created zero-filled on first use).  For a 3x3 conv
    weight [Cout,Cin,3,3] the buffer is channels_last and the returned tensor is its contiguous [Cout,3,3,Cin] view."""
    g = p.grad
    if g is None:
        if conv3x3:
            g = torch.zeros((p.shape[0], 3, 3, p.shape[1]), dtype=torch.float32, device=p.device).permute(0, 3, 1, 2)
        else:
            g = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
        p.grad = g
    if g.dtype != torch.float32:
        raise TypeError("hcp_diffusion_amd: host gradients accumulate in fp32 (fp32 master parameters expected)")
    if conv3x3:
        g = g.permute(0, 2, 3, 1)
    if not g.is_contiguous():
        raise RuntimeError("hcp_diffusion_amd: .grad of a trainable host parameter must be dense (3x3 conv weights: channels_last)")
    return g


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ T (alpha Bu)^T) + bias (+ residual);  T = x Ad^T.   Reference arithmetic:
    LoraPatchContainer.forward / LoraBlock.post_forward (lora_base_patch.py:20-35,68-74)."""

    @staticmethod
    def forward(ctx, x, residual, w_down, w_up, host, lora, out_f32=False, hw=None, hb=None, residual_lo=None, stream=False, geglu=False):
        """stream: the residual is a (hi | lo) residual stream (residual, residual_lo — lo may be None where the stream starts) and the
        result is the pair (y_hi, y_lo); their gradients come back as a pair too and pass to the residual inputs as they are.
        geglu: the layer is diffusers' GEGLU projection — returns ((h | g), bf16(h * gelu(g))), the second formed in the GEMM epilogue
        from the fp32 values (K.gemm(want_gact)); it carries no gradient of its own: _GegluLinearFn, which consumes both, sends the
        whole gradient back through (h | g)."""
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        res2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        lo2 = residual_lo.reshape(-1, residual_lo.shape[-1]) if residual_lo is not None else None
        pk = host.packed()
        T = None
        if lora is not None and getattr(lora, "wide", False):      # rank > 32: skinny side GEMM + K-extension
            lp = lora.packed()
            T = K.gemm(x2, lp.ad)
            y = K.gemm(x2, pk.w, a2=T, b2=lp.bu, bias=pk.bias, residual=res2, residual_lo=lo2, want_lo=stream, want_gact=geglu)
        elif lora is not None:
            lp = lora.packed()
            y, T = K.gemm_lora(x2, pk.w, lp.ad, lp.bu, bias=pk.bias, residual=res2, residual_lo=lo2, want_lo=stream, want_gact=geglu)
        else:
            y = K.gemm(x2, pk.w, bias=pk.bias, residual=res2, out_f32=out_f32, residual_lo=lo2, want_lo=stream, want_gact=geglu)
        ctx.host, ctx.lora = host, lora
        ctx.wg = current_wgrad()
        ctx.train_w, ctx.train_b = hw is not None, hb is not None
        ctx.save_for_backward(x2 if (lora is not None or hw is not None) else None, T)
        ctx.xshape = shp
        ctx.has_res = residual is not None
        ctx.stream, ctx.has_lo = stream, residual_lo is not None
        if geglu:
            act = y[1].view(*shp[:-1], y[1].shape[-1])
            ctx.mark_non_differentiable(act)
            ctx.set_materialize_grads(False)           # (else autograd fills a zero "gradient" of act for every backward: 16 launches per step)
            return y[0].view(*shp[:-1], y[0].shape[-1]), act
        if stream:
            ctx.set_materialize_grads(False)           # the lo image of the last block has no consumer: its gradient stays None
            return y[0].view(*shp[:-1], y[0].shape[-1]), y[1].view(*shp[:-1], y[1].shape[-1])
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy, dy_lo=None):
        if dy is None:                                 # (stream mode, nothing downstream took the gradient)
            assert dy_lo is None
            return (None,) * 12
        x2, T = ctx.saved_tensors
        host, lora = ctx.host, ctx.lora
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        pk = host.packed()
        dx = None
        if lora is not None and getattr(lora, "wide", False):
            lp = lora.packed()
            U = K.gemm(dy2, lp.but)                    # dY (alpha W_up): [M, Rp]
            if ctx.needs_input_grad[0]:
                dx = K.gemm(dy2, pk.wt, a2=U, b2=lp.wdt)
            for blk, s0 in lora.members():             # one wide block, or several blocks sharing the wide slot range (lora.MultiLora)
                gd, gu = blk.grad_views()
                for j in range(0, blk.rank, 32):       # 32 rank columns per launch of the skinny reduction kernel
                    pj = min(32, blk.rank - j)
                    K.lora_wgrad(U[:, s0 + j:s0 + j + 32], x2, gd[j:j + pj], pj, 1.0, False)                # dW_down = U^T x
                    K.lora_wgrad(T[:, s0 + j:s0 + j + 32], dy2, gu, pj, blk.alpha_f, True, out_col0=j)    # dW_up = alpha dY^T T
        elif lora is not None:
            lp = lora.packed()
            if ctx.needs_input_grad[0]:
                dx, U = K.gemm_lora(dy2, pk.wt, lp.but, lp.adt)
            else:
                U = K.gemm(dy2, lp.but)
            for blk, s0 in lora.members():             # one block, or several sharing the 32 rank slots (lora.MultiLora)
                gd, gu = blk.grad_views()
                ctx.wg.add(U, x2, gd, T, dy2, gu, blk.rank, blk.alpha_f, s0, K.t_lo(U), K.t_lo(T))
        elif ctx.needs_input_grad[0]:
            dx = K.gemm(dy2, pk.wt)
        if ctx.train_w:                                # dW[N,K] += dY^T X (nn.Linear [N,K]; 1x1 conv [N,K,1,1] = same memory)
            gw = grad_buffer(host.weight)
            K.wgrad_linear(dy2, x2, gw.view(gw.shape[0], -1))
        if ctx.train_b:
            K.colsum(dy2, grad_buffer(host.bias))
        if dx is not None:
            dx = dx.view(ctx.xshape)
        # the stream's gradient is a (hi | lo) pair as well: the GEMMs above read its hi image (the reference's autograd casts the fp32
        # stream gradient to bf16 in front of the same mm), the residual path hands both images on untouched
        return dx, (dy if ctx.has_res else None), None, None, None, None, None, None, None, (dy_lo if ctx.has_lo else None), None, None


def linear_geglu(x, host, lora=None):
    """((h | g), bf16(h * gelu(g))) of diffusers' GEGLU projection in one launch; hand both to geglu_linear."""
    wd = lora.layer.W_down if lora is not None else None
    wu = lora.layer.W_up if lora is not None else None
    return _LinearFn.apply(x, None, wd, wu, host, lora, False, _tr(host.weight), _tr(host.bias), None, False, True)


def linear(x, host, lora=None, residual=None, out_f32=False):
    wd = lora.layer.W_down if lora is not None else None
    wu = lora.layer.W_up if lora is not None else None
    hw, hb = _tr(host.weight), _tr(host.bias)
    if out_f32 and (lora is not None or x.requires_grad or hw is not None or hb is not None):
        raise NotImplementedError("hcp_diffusion_amd: fp32 linear output is only provided for the gradient-free time-embedding path")
    return _LinearFn.apply(x, residual, wd, wu, host, lora, out_f32, hw, hb)


def linear_stream(x, host, lora, hi, lo):
    """(y_hi, y_lo) = split(x W^T [+ LoRA] + bias + hi + lo): a Linear whose residual is a (hi | lo) residual stream (lo: None where the
    stream starts).  See GemmParams::residual_lo (csrc/gemm_params.h)."""
    wd = lora.layer.W_down if lora is not None else None
    wu = lora.layer.W_up if lora is not None else None
    return _LinearFn.apply(x, hi, wd, wu, host, lora, False, _tr(host.weight), _tr(host.bias), lo, True)


class _LinearGroupFn(torch.autograd.Function):
    """Several frozen Linear layers (+ their LoRA blocks) reading the same input, as ONE fused-LoRA GEMM:
    y = x [W_0; W_1; ...]^T + T E^T with every block's rank slots packed side by side (lora.FusedLoraGroup)."""

    @staticmethod
    def forward(ctx, x, group, *lora_params):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        w, _ = group.packed_host()
        T = None
        if group.has_lora:
            o = group.bucket.packed_group(group)
            y, T = K.gemm_lora(x2, w, o.ad, o.bu)
        else:
            y = K.gemm(x2, w)
        ctx.group = group
        ctx.wg = current_wgrad()
        ctx.save_for_backward(x2 if group.has_lora else None, T)
        ctx.xshape = shp
        return y.view(*shp[:-1], group.n_total)

    @staticmethod
    def backward(ctx, dy):
        x2, T = ctx.saved_tensors
        g = ctx.group
        dy2 = dy.reshape(-1, g.n_total)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        _, wt = g.packed_host()
        dx = None
        if g.has_lora:
            o = g.bucket.packed_group(g)
            if ctx.needs_input_grad[0]:
                dx, U = K.gemm_lora(dy2, wt, o.but, o.adt)
            else:
                U = K.gemm(dy2, o.but)
            for blk, n0, s0, host, sc_ in zip(g.blocks, g.n_off, g.slot_off, g.hosts, g.out_scale):
                if blk is not None:
                    gd, gu = blk.grad_views()
                    ctx.wg.add(U, x2, gd, T, dy2[:, n0:n0 + host.weight.shape[0]], gu, blk.rank, blk.alpha_f * sc_, s0, K.t_lo(U), K.t_lo(T))
        elif ctx.needs_input_grad[0]:
            dx = K.gemm(dy2, wt)
        if dx is not None:
            dx = dx.view(ctx.xshape)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _CtxKVFn(torch.autograd.Function):
    """K/V projections of every cross-attention layer from the shared prompt states in three launches (lora.CtxBatch).  Outputs: one
    [.., N_g] column-slice VIEW of the joint result per group; their gradients arrive in the matching slices of one joint buffer
    (the attention backward writes them there: `_hcp_grad_buf`), the input needs no gradient by construction."""

    @staticmethod
    def forward(ctx, x, batch, *lora_params):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        ad_all, b_cat = batch.operands()
        T_all = None
        if ad_all is not None and batch.split:
            T_all = K.split_hi_lo(K.gemm(x2, ad_all, out_f32=True))  # [M, 2 * 32 G] = (hi | lo) of every layer's x W_down^T (fp32 accumulator)
            y = K.gemm(K.concat_channels(x2, T_all), b_cat)          # [M, sum N_g]; b_cat = [W_all | BU | BU]
        elif ad_all is not None:
            T_all = K.gemm(x2, ad_all)                               # [M, 32 G]: every layer's x W_down^T
            y = K.gemm(K.concat_channels(x2, T_all), b_cat)          # [M, sum N_g]
        else:
            y = K.gemm(x2, b_cat)
        ctx.batch, ctx.wg = batch, current_wgrad()
        ctx.dkv_all = torch.empty_like(y)
        ctx.save_for_backward(x2, T_all)
        y3 = y.view(*shp[:-1], batch.n_total)
        return tuple(y3[..., o:o + g.n_total] for g, o in zip(batch.groups, batch.n_off))

    @staticmethod
    def backward(ctx, *dkvs):
        x2, T_all = ctx.saved_tensors
        batch, dall = ctx.batch, ctx.dkv_all
        joint = T_all is not None and batch.all_have_lora() and all(d is not None for d in dkvs)
        for g, off, d in zip(batch.groups, batch.n_off, dkvs):
            if d is None or not g.has_lora:
                continue
            sl = dall[:, off:off + g.n_total]
            d2 = d.reshape(-1, g.n_total) if d.dim() != 2 else d
            if d2.data_ptr() != sl.data_ptr() or d2.stride() != sl.stride():
                sl.copy_(d2)                                          # (a consumer that did not write in place)
        U_all = None
        if joint:                                                     # every layer's U = dY W_up in ONE deep-K GEMM (block-diagonal operand)
            batch.operands()
            U_all = K.split_hi_lo(K.gemm(dall, batch.but_all, out_f32=True)) if batch.split else K.gemm(dall, batch.but_all)   # [M, 32 G] (x 2: hi | lo)
        for gi, (g, off, d) in enumerate(zip(batch.groups, batch.n_off, dkvs)):
            if d is None or not g.has_lora:
                continue
            sl = dall[:, off:off + g.n_total]
            U = U_all[:, 32 * gi:32 * gi + 32] if joint else K.gemm(sl, g.bucket.packed_group(g).but)
            T = T_all[:, 32 * gi:32 * gi + 32]
            for blk, n0, s0, host, sc_ in zip(g.blocks, g.n_off, g.slot_off, g.hosts, g.out_scale):
                if blk is not None:
                    gd, gu = blk.grad_views()
                    ctx.wg.add(U, x2, gd, T, sl[:, n0:n0 + host.weight.shape[0]], gu, blk.rank, blk.alpha_f * sc_, s0,
                               batch.k2 if (joint and batch.split) else 0, batch.k2 if batch.split else 0)
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def ctx_kv(x, batch):
    """[(k|v) of group g as a view [.., N_g]] for every group of a lora.CtxBatch; x = the prompt states (no gradient wanted)."""
    assert not x.requires_grad
    params = [p for g in batch.groups for b in g.blocks if b is not None for p in (b.layer.W_down, b.layer.W_up)]
    outs = _CtxKVFn.apply(x, batch, *params)
    node = outs[0].grad_fn
    if node is not None:                                             # hand every consumer its slice of the joint gradient buffer
        d3 = node.dkv_all.view(*x.shape[:-1], batch.n_total)
        for kv, g, o in zip(outs, batch.groups, batch.n_off):
            kv._hcp_grad_buf = d3[..., o:o + g.n_total]
    return outs


def linear_group(x, group):
    for h in group.hosts:                              # unet._fusable_linear never groups trainable hosts
        assert _tr(h.weight) is None and _tr(h.bias) is None, "fused projection groups require frozen host weights"
    params = [p for b in group.blocks if b is not None for p in (b.layer.W_down, b.layer.W_up)]
    return _LinearGroupFn.apply(x, group, *params)


class _Conv3x3Fn(torch.autograd.Function):
    """3x3/pad-1 conv on NHWC bf16 (+bias, + per-sample row bias (time embedding), + residual); optional second
    input (channel concat), stride 2, fused nearest-2x upsample."""

    @staticmethod
    def forward(ctx, x1, x2, rowbias, residual, host, stride, upsample, hw=None, hb=None, lora=None, w_down=None, w_up=None):
        pk = host.packed()
        T = None
        if lora is not None and getattr(lora, "wide", False):      # rank > 32: the side path's product joins through one more GEMM
            lp = lora.packed()
            rp = lora.rank_pad
            T = K.conv3x3(x1, lp.ad, rp, x2=x2, stride=stride, upsample=upsample)
            y0 = K.conv3x3(x1, pk.w, pk.cout, x2=x2, stride=stride, upsample=upsample, bias=pk.bias, rowbias=rowbias, residual=residual)
            y = K.gemm(T.view(-1, rp), lp.bu, residual=y0.view(-1, pk.cout)).view_as(y0)
        elif lora is not None:     # LoCon side path: T = conv3x3(x, W_down) [.,32]; y = conv(x, W) + T (alpha W_up)^T as a K-extension
            lp = lora.packed()
            T = K.conv3x3(x1, lp.ad, 32, x2=x2, stride=stride, upsample=upsample)
            y = K.conv3x3(x1, pk.w, pk.cout, x2=x2, stride=stride, upsample=upsample, bias=pk.bias, rowbias=rowbias, residual=residual,
                          a2=T, b2=lp.bu)
        else:
            y = K.conv3x3(x1, pk.w, pk.cout, x2=x2, stride=stride, upsample=upsample, bias=pk.bias, rowbias=rowbias, residual=residual)
        ctx.host, ctx.stride, ctx.upsample, ctx.lora = host, stride, upsample, lora
        ctx.in_shape = x1.shape
        ctx.c2 = x2.shape[-1] if x2 is not None else 0
        ctx.has_res = residual is not None
        ctx.train_w, ctx.train_b = hw is not None, hb is not None
        keep_x = hw is not None or lora is not None
        ctx.save_for_backward(x1 if keep_x else None, x2 if keep_x else None, T)
        return y

    @staticmethod
    def backward(ctx, dy):
        pk = ctx.host.packed()
        dy = dy.contiguous()
        B, H, W, C1 = ctx.in_shape
        hw = (H * 2, W * 2) if ctx.upsample else (H, W)
        dx1 = dx2 = None
        x1, x2, T = ctx.saved_tensors
        lora = ctx.lora
        dl1 = dl2 = None
        if lora is not None:       # U = dY (alpha W_up) [.,32]; dW_up = alpha dY^T T; dW_down = U^T im2col(x); dX += dgrad(U, W_down)
            lp = lora.packed()
            rp = lp.but.shape[0]                       # 32 rank slots, or the padded rank of a wide block
            U = K.gemm(dy.view(-1, dy.shape[-1]), lp.but).view(*dy.shape[:-1], rp)
            T2 = T.view(-1, rp)
            for blk, s0 in lora.members():             # one block, or several sharing the 32 rank slots (lora.MultiLora)
                gd, gu = blk.grad_views()
                for j in range(0, blk.rank, 32):       # dW_up = alpha dY^T T: 32 rank columns per launch of the skinny reduction kernel
                    K.lora_wgrad(T2[:, s0 + j:], dy.view(-1, dy.shape[-1]), gu, min(32, blk.rank - j), blk.alpha_f, True, out_col0=j)
                K.wgrad_conv3x3(U, x1, gd, x2=x2, stride=ctx.stride, upsample=ctx.upsample, cout=blk.rank, col0=s0)
            if ctx.needs_input_grad[0]:
                dl1 = K.conv3x3(U, lp.wdl[:C1], C1, mode=1, stride=ctx.stride, out_hw=hw)
            if ctx.c2 and ctx.needs_input_grad[1]:
                dl2 = K.conv3x3(U, lp.wdl[C1:], ctx.c2, mode=1, stride=ctx.stride, out_hw=hw)
        if ctx.needs_input_grad[0]:
            dx1 = K.conv3x3(dy, pk.wd[:C1], C1, mode=1, stride=ctx.stride, out_hw=hw, residual=dl1)
            if ctx.upsample:
                dx1 = K.upsample2x_bwd(dx1)
        if ctx.c2 and ctx.needs_input_grad[1]:
            dx2 = K.conv3x3(dy, pk.wd[C1:], ctx.c2, mode=1, stride=ctx.stride, out_hw=hw, residual=dl2)
        drb = None
        dy2 = dy.view(-1, dy.shape[-1])
        if ctx.needs_input_grad[2]:                    # per-sample row bias (time embedding): sum over the sample's pixels
            drb = torch.zeros((B, dy.shape[-1]), dtype=torch.float32, device=dy.device)
            K.colsum(dy2, drb, dy.shape[1] * dy.shape[2])
        if ctx.train_w:
            K.wgrad_conv3x3(dy, x1, grad_buffer(ctx.host.weight, True), x2=x2, stride=ctx.stride, upsample=ctx.upsample)
        if ctx.train_b:
            K.colsum(dy2, grad_buffer(ctx.host.bias))
        return dx1, dx2, drb, (dy if ctx.has_res else None), None, None, None, None, None, None, None, None


def conv3x3(x1, host, *, x2=None, rowbias=None, residual=None, stride=1, upsample=False, lora=None):
    wd = lora.layer.W_down if lora is not None else None
    wu = lora.layer.W_up if lora is not None else None
    return _Conv3x3Fn.apply(x1, x2, rowbias, residual, host, stride, upsample, _tr(host.weight), _tr(host.bias), lora, wd, wu)


def _gn_affine(ctx, x, dy, g, b, stats):
    if ctx.train:
        K.groupnorm_affine_grad(x, dy, g, b, stats, ctx.gn.num_groups, ctx.silu, grad_buffer(ctx.gn.weight), grad_buffer(ctx.gn.bias))


def _norm_tr(m):
    w, b = _tr(m.weight), _tr(m.bias)
    if (w is None) != (b is None):
        raise NotImplementedError("hcp_diffusion_amd: a norm layer's weight and bias must be trainable together")
    return w, b


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gn, silu, hw=None, hb=None):
        g, b = gn.f32_params()
        y, stats = K.groupnorm_fwd(x, g, b, gn.num_groups, gn.eps, silu)
        ctx.save_for_backward(x, stats)
        ctx.gn, ctx.silu, ctx.train = gn, silu, hw is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        g, b = ctx.gn.f32_params()
        dy = dy.contiguous()
        _gn_affine(ctx, x, dy, g, b, stats)
        dx = K.groupnorm_bwd(x, dy, g, b, stats, ctx.gn.num_groups, ctx.silu) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None


def groupnorm(x, gn, silu):
    return _GroupNormFn.apply(x, gn, silu, *_norm_tr(gn))


class _GroupNormForkFn(torch.autograd.Function):
    """(norm(x), x): the block input forks into the normalised branch and the residual path.  Backward receives both
    gradients at once and adds the residual-path one inside the norm-backward kernel (no separate accumulation)."""

    @staticmethod
    def forward(ctx, x, gn, silu, hw=None, hb=None):
        g, b = gn.f32_params()
        y, stats = K.groupnorm_fwd(x, g, b, gn.num_groups, gn.eps, silu)
        ctx.save_for_backward(x, stats)
        ctx.gn, ctx.silu, ctx.train = gn, silu, hw is not None
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, stats = ctx.saved_tensors
        g, b = ctx.gn.f32_params()
        if dy is None:
            return dskip, None, None, None, None
        dy = dy.contiguous()
        _gn_affine(ctx, x, dy, g, b, stats)
        return K.groupnorm_bwd(x, dy, g, b, stats, ctx.gn.num_groups, ctx.silu,
                               dskip.contiguous() if dskip is not None else None), None, None, None, None


def groupnorm_fork(x, gn, silu):
    return _GroupNormForkFn.apply(x, gn, silu, *_norm_tr(gn))


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln, hw=None, hb=None):
        g, b = ln.f32_params()
        y, stats = K.layernorm_fwd(x, g, b, ln.eps)
        ctx.save_for_backward(x, stats)
        ctx.ln, ctx.train = ln, hw is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        g, _ = ctx.ln.f32_params()
        dy = dy.contiguous()
        if ctx.train:
            K.layernorm_affine_grad(x, dy, stats, grad_buffer(ctx.ln.weight), grad_buffer(ctx.ln.bias))
        return (K.layernorm_bwd(x, dy, g, stats) if ctx.needs_input_grad[0] else None), None, None, None


def layernorm(x, ln):
    return _LayerNormFn.apply(x, ln, *_norm_tr(ln))


class _LayerNormForkFn(torch.autograd.Function):
    """(layer_norm(x), x) for pre-norm residual blocks; see _GroupNormForkFn."""

    @staticmethod
    def forward(ctx, x, ln, hw=None, hb=None, x_lo=None, stream=False):
        """stream: x is the hi image of a (hi | lo) residual stream, x_lo its lo image (None where the stream starts); returns
        (LN(x + x_lo), x, x_lo) and takes the skip gradient back as a pair."""
        g, b = ln.f32_params()
        y, stats = K.layernorm_fwd(x, g, b, ln.eps, x_lo=x_lo)
        ctx.save_for_backward(x, stats, x_lo)
        ctx.ln, ctx.train, ctx.stream = ln, hw is not None, stream
        if stream:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x), (x_lo.view_as(x_lo) if x_lo is not None else None)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip, dskip_lo=None):
        x, stats, x_lo = ctx.saved_tensors
        g, _ = ctx.ln.f32_params()
        if dy is None:
            return dskip, None, None, None, dskip_lo, None
        dy = dy.contiguous()
        if ctx.train:
            K.layernorm_affine_grad(x, dy, stats, grad_buffer(ctx.ln.weight), grad_buffer(ctx.ln.bias))
        if ctx.stream:
            want_lo = x_lo is not None                 # the stream's first norm hands ONE bf16 gradient back to the producer of x
            r = K.layernorm_bwd(x, dy, g, stats, dskip.contiguous() if dskip is not None else None, x_lo=x_lo,
                                addend_lo=dskip_lo.contiguous() if (dskip_lo is not None and dskip is not None) else None, want_lo=want_lo)
            return (r[0], None, None, None, r[1], None) if want_lo else (r, None, None, None, None, None)
        return K.layernorm_bwd(x, dy, g, stats, dskip.contiguous() if dskip is not None else None), None, None, None, None, None


def layernorm_fork(x, ln):
    return _LayerNormForkFn.apply(x, ln, *_norm_tr(ln))


def layernorm_fork_stream(hi, lo, ln):
    """(LN(hi + lo), hi, lo) on a (hi | lo) residual stream; lo may be None (the stream's first norm)."""
    return _LayerNormForkFn.apply(hi, ln, *_norm_tr(ln), lo, True)


class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return K.geglu_fwd(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return K.geglu_bwd(h, dy.contiguous())


geglu = _GegluFn.apply


class _GegluLinearFn(torch.autograd.Function):
    """y = FFout(h * gelu(g)) (+ residual) for (h | g) = hg, i.e. diffusers' FeedForward after its first projection
    (GEGLU + Dropout(0) + Linear, cfgs/unet_struct.txt:27-33) as ONE autograd node: the forward is the GEGLU kernel + the (fused-LoRA) GEMM
    as before; the backward runs the input-gradient GEMM with the GEGLU backward in its epilogue (hcp_gemm_geglu_bwd_bf16) — the
    [M, 4C] gradient of the GEGLU output is never written or re-read, and the stand-alone geglu_bwd launch is gone.  Same arithmetic as
    _GegluFn + _LinearFn (the epilogue rounds dY_ff to bf16 before the two products, as the two-kernel form does)."""

    @staticmethod
    def forward(ctx, hg, residual, w_down, w_up, host, lora, hw=None, hb=None, residual_lo=None, stream=False, gact=None):
        """gact: bf16(h * gelu(g)) already formed by the projection's epilogue (linear_geglu), else the stand-alone pass runs here."""
        shp = hg.shape
        hg2 = hg.reshape(-1, shp[-1])
        x2 = gact.reshape(-1, gact.shape[-1]) if gact is not None else K.geglu_fwd(hg2)
        res2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        lo2 = residual_lo.reshape(-1, residual_lo.shape[-1]) if residual_lo is not None else None
        pk = host.packed()
        T = None
        if lora is not None:
            lp = lora.packed()
            y, T = K.gemm_lora(x2, pk.w, lp.ad, lp.bu, bias=pk.bias, residual=res2, residual_lo=lo2, want_lo=stream)
        else:
            y = K.gemm(x2, pk.w, bias=pk.bias, residual=res2, residual_lo=lo2, want_lo=stream)
        ctx.host, ctx.lora = host, lora
        ctx.wg = current_wgrad()
        ctx.train_w, ctx.train_b = hw is not None, hb is not None
        ctx.save_for_backward(hg2, x2 if (lora is not None or hw is not None) else None, T)
        ctx.hshape = shp
        ctx.has_res = residual is not None
        ctx.has_lo = residual_lo is not None
        if stream:                                     # (hi | lo) residual stream: see _LinearFn
            ctx.set_materialize_grads(False)
            return y[0].view(*shp[:-1], y[0].shape[-1]), y[1].view(*shp[:-1], y[1].shape[-1])
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy, dy_lo=None):
        if dy is None:
            assert dy_lo is None
            return (None,) * 11
        hg2, x2, T = ctx.saved_tensors
        host, lora = ctx.host, ctx.lora
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        pk = host.packed()
        dhg = None
        if lora is not None:
            lp = lora.packed()
            if ctx.needs_input_grad[0]:
                dhg, U = K.gemm_geglu_bwd(dy2, pk.wt, hg2, l=lp.but, e=lp.adt)
            else:
                U = K.gemm(dy2, lp.but)
            for blk, s0 in lora.members():
                gd, gu = blk.grad_views()
                ctx.wg.add(U, x2, gd, T, dy2, gu, blk.rank, blk.alpha_f, s0, K.t_lo(U), K.t_lo(T))
        elif ctx.needs_input_grad[0]:
            dhg, _ = K.gemm_geglu_bwd(dy2, pk.wt, hg2)
        if ctx.train_w:
            gw = grad_buffer(host.weight)
            K.wgrad_linear(dy2, x2, gw.view(gw.shape[0], -1))
        if ctx.train_b:
            K.colsum(dy2, grad_buffer(host.bias))
        if dhg is not None:
            dhg = dhg.view(ctx.hshape)
        return dhg, (dy if ctx.has_res else None), None, None, None, None, None, None, (dy_lo if ctx.has_lo else None), None, None


def geglu_linear(hg, host, lora=None, residual=None, gact=None):
    """residual: a tensor, or the (hi, lo) pair of a (hi | lo) residual stream — the result is then a pair too.
    gact: the second output of linear_geglu (the GEGLU product from the projection's own epilogue)."""
    wd = lora.layer.W_down if lora is not None else None
    wu = lora.layer.W_up if lora is not None else None
    if isinstance(residual, tuple):
        return _GegluLinearFn.apply(hg, residual[0], wd, wu, host, lora, _tr(host.weight), _tr(host.bias), residual[1], True, gact)
    return _GegluLinearFn.apply(hg, residual, wd, wu, host, lora, _tr(host.weight), _tr(host.bias), None, False, gact)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, key_bias=None, causal=False):
        o, lse = K.attention_fwd(q, k, v, heads, key_bias=key_bias, causal=causal)
        ctx.save_for_backward(q, k, v, o, lse, key_bias)
        ctx.heads, ctx.causal = heads, causal
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_bias = ctx.saved_tensors
        dq, dk, dv = K.attention_bwd(q.contiguous(), k.contiguous(), v.contiguous(), o, do.contiguous(), lse, ctx.heads,
                                     key_bias=key_bias, causal=ctx.causal)
        return dq, dk, dv, None, None, None


def attention(q, k, v, heads, key_bias=None, causal=False):
    """key_bias: optional fp32 [B,Nk] additive key mask (constant: no gradient); causal: the text encoder's triangular mask."""
    return _AttentionFn.apply(q, k, v, heads, key_bias, causal)


class _QuickGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.quick_gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.quick_gelu(x, dy.contiguous())


def quick_gelu(x):
    return _QuickGeluFn.apply(x)


class _AttentionPackedFn(torch.autograd.Function):
    """Attention on fused projection buffers: `a` = [B,N,3C] (q|k|v, self-attention) or `a` = q [B,N,C] with
    `kv` = [B,Nk,2C] (k|v, cross-attention).  The kernels read the column slices in place and write the gradients
    into ONE buffer per input — no slice/concat nodes in the autograd graph."""

    @staticmethod
    def forward(ctx, a, kv, heads, key_bias=None, q_prescaled=False):
        if kv is None:
            C = a.shape[-1] // 3
            q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
        else:
            C = a.shape[-1]
            q, k, v = a, kv[..., :C], kv[..., C:]
        o, lse = K.attention_fwd(q, k, v, heads, key_bias=key_bias, q_prescaled=q_prescaled)
        ctx.save_for_backward(a, kv, o, lse, key_bias)
        ctx.heads, ctx.C, ctx.pre = heads, C, q_prescaled
        ctx.dkv_buf = getattr(kv, "_hcp_grad_buf", None)      # a slice of ops._CtxKVFn's joint gradient buffer (same strides as kv)
        return o

    @staticmethod
    def backward(ctx, do):
        a, kv, o, lse, key_bias = ctx.saved_tensors
        C = ctx.C
        if kv is None:
            da = torch.empty_like(a)
            q, k, v = a[..., :C], a[..., C:2 * C], a[..., 2 * C:]
            K.attention_bwd(q, k, v, o, do.contiguous(), lse, ctx.heads, out=(da[..., :C], da[..., C:2 * C], da[..., 2 * C:]),
                            key_bias=key_bias, q_prescaled=ctx.pre)
            return da, None, None, None, None
        da = torch.empty_like(a)
        dkv = ctx.dkv_buf if ctx.dkv_buf is not None else torch.empty_like(kv)
        if dkv.stride() != kv.stride():                       # (a strided kv whose producer offered no buffer)
            dkv = torch.empty_strided(kv.shape, kv.stride(), dtype=kv.dtype, device=kv.device)
        K.attention_bwd(a, kv[..., :C], kv[..., C:], o, do.contiguous(), lse, ctx.heads, out=(da, dkv[..., :C], dkv[..., C:]),
                        key_bias=key_bias, q_prescaled=ctx.pre)
        return da, dkv, None, None, None


def attention_packed(a, kv, heads, key_bias=None, q_prescaled=False):
    """q_prescaled: the q columns of `a` were produced by a projection group with out_scale = d^-0.5 * log2(e) (lora.FusedLoraGroup):
    the kernels skip the per-score multiply and return the gradient w.r.t. the scaled tensor, which is what that group's backward expects."""
    return _AttentionPackedFn.apply(a, kv, heads, key_bias, q_prescaled)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


add = _AddFn.apply


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.c1 = a.shape[-1]
        return K.concat_channels(a, b)

    @staticmethod
    def backward(ctx, d):
        return K.split_channels(d.contiguous(), ctx.c1)


concat_channels = _ConcatFn.apply


class _MergedLoraFn(torch.autograd.Function):
    """LoRA on a host the side-path kernels do not serve (conv_in / conv_out: 4 latent channels): the reference's own arithmetic,
    ``layer(x, W + sum_b alpha_b W_up_b W_down_b)`` (LoraPatchContainer.forward, lora_base_patch.py:20-35).  The host's forward runs on a
    shadow of the host module whose weight is the merged FP32 tensor as a trainable leaf (whatever the host's own dtype: a frozen bf16
    host keeps its fp32 gradient path, and W + dW is rounded once, by the operand pack), so the host's own weight-gradient kernel produces
    dW_eff; the factor gradients follow from it (dW_down = alpha W_up^T dW_eff, dW_up = alpha dW_eff W_down^T: two small contractions on
    tensors of the layer's weight size) and go into the blocks' bucket views; a TRAINABLE host weight receives dW_eff as well (the
    reference differentiates layer(x, host_weight + weight): both terms get the gradient).
    Memory: W_eff and its gradient are two fp32 tensors of the host weight's size per call (a 1280 x 2560 x 3 x 3 host: 118 MB each)."""

    @staticmethod
    def merged_weight(host, blocks):
        w_eff = host.weight.detach().to(torch.float32, copy=True)        # keeps the 3x3 weight's channels_last storage
        for b in blocks:
            w_eff += b.alpha_f * torch.einsum("or,rikl->oikl", b.layer.W_up.detach()[:, :, 0, 0].float(), b.layer.W_down.detach().float())
        return w_eff

    @staticmethod
    def shadow_of(host, w_eff, trainable):
        import copy
        shadow = copy.copy(host)
        shadow._parameters = dict(host._parameters)
        shadow._parameters["weight"] = torch.nn.Parameter(w_eff, requires_grad=trainable)
        shadow._pk = None
        return shadow

    @staticmethod
    def forward(ctx, x, residual, x2, rowbias, host, blocks, upsample, hw, *factors):
        shadow = _MergedLoraFn.shadow_of(host, _MergedLoraFn.merged_weight(host, blocks), True)
        p = shadow._parameters["weight"]
        def leaf(t):                                           # the inner graph must not reach back into the outer one
            if t is None:
                return None
            u = t.detach()
            return u.requires_grad_(True) if t.requires_grad else u
        with torch.enable_grad():
            xin, rin, x2in, rbin = leaf(x), leaf(residual), leaf(x2), leaf(rowbias)
            kw = {k: v for k, v in (("residual", rin), ("x2", x2in), ("rowbias", rbin)) if v is not None}
            if upsample:
                kw["upsample"] = True
            y = shadow.forward(xin, **kw)                       # (conv_in / conv_out take the input alone)
        ctx.inner = (xin, rin, x2in, rbin, y, p)
        ctx.blocks, ctx.host, ctx.train_w = blocks, host, hw is not None
        return y.detach()

    @staticmethod
    def backward(ctx, dy):
        xin, rin, x2in, rbin, y, p = ctx.inner
        torch.autograd.backward(y, dy)
        g = p.grad
        for b in ctx.blocks:
            b.grad_views()                                     # (re-)attaches the factors' .grad views of the bucket
            wu = b.layer.W_up.detach()[:, :, 0, 0].float()
            b.layer.W_down.grad.add_(b.alpha_f * torch.einsum("or,oikl->rikl", wu, g))
            b.layer.W_up.grad.add_((b.alpha_f * torch.einsum("oikl,rikl->or", g, b.layer.W_down.detach().float()))[:, :, None, None])
        if ctx.train_w:                                        # full fine-tune + LoRA on the same host: dW_host = dW_eff
            hw = ctx.host.weight
            if hw.dim() == 4 and tuple(hw.shape[2:]) == (3, 3):
                grad_buffer(hw, conv3x3=True).add_(g.permute(0, 2, 3, 1))
            else:
                grad_buffer(hw).add_(g.view_as(hw))
        ctx.inner = None
        gr = lambda t: t.grad if (t is not None and t.requires_grad) else None
        return gr(xin), gr(rin), gr(x2in), gr(rbin), None, None, None, None, *([None] * len(ctx.blocks) * 2)


_MERGED_EVAL_CACHE = {}   # id(host) -> (key, shadow): gradient-free calls (the sampler: one per denoising step) reuse W_eff and its operand pack
_MERGED_GEN = [0]


def invalidate_merged_cache():
    """Called by everything that writes trained parameters WITHOUT moving a torch version counter: the fused clip + AdamW
    kernel takes raw pointers, a captured optimizer step replays without running any Python — NativeTrainer bumps this after every
    optimizer step (eager or replayed), LoraBucket.pack() / HostBucket.repack() whenever they re-derive operands (loads, EMA swaps)."""
    _MERGED_GEN[0] += 1


def merged_lora_call(host, blocks, x, residual=None, x2=None, rowbias=None, upsample=False):
    factors = [p for b in blocks for p in (b.layer.W_down, b.layer.W_up)]
    needs = torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in (x, residual, x2, rowbias)) or
                                          any(p.requires_grad for p in factors) or host.weight.requires_grad or
                                          (host.bias is not None and host.bias.requires_grad))
    if not needs:                                              # no inner autograd graph, no re-merge / re-pack while nothing changed
        key = ((_MERGED_GEN[0], host.weight._version, host.weight.data_ptr()) + tuple((p._version, p.data_ptr()) for p in factors) +
               tuple(b.alpha_f for b in blocks))
        hit = _MERGED_EVAL_CACHE.get(id(host))
        if hit is None or hit[0] != key or hit[2]() is not host:
            import weakref
            if len(_MERGED_EVAL_CACHE) >= 8:
                _MERGED_EVAL_CACHE.pop(next(iter(_MERGED_EVAL_CACHE)))
            hit = (key, _MergedLoraFn.shadow_of(host, _MergedLoraFn.merged_weight(host, blocks), False), weakref.ref(host))
            _MERGED_EVAL_CACHE[id(host)] = hit
        kw = {k: v for k, v in (("residual", residual), ("x2", x2), ("rowbias", rowbias)) if v is not None}
        if upsample:
            kw["upsample"] = True
        with torch.no_grad():
            return hit[1].forward(x, **kw)
    return _MergedLoraFn.apply(x, residual, x2, rowbias, host, blocks, upsample, _tr(host.weight), *factors)


class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.silu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.silu_bwd(x, dy.contiguous())


silu = _SiluFn.apply


class _ConvOutFn(torch.autograd.Function):
    """conv_out: NHWC bf16 -> 3x3 conv (fp32 accum/out) -> NCHW fp32 `.sample`.  Backward: NCHW fp32 gradient ->
    channel-padded NHWC bf16 -> data-gradient conv."""

    @staticmethod
    def forward(ctx, x, host, hw=None, hb=None):
        pk = host.packed()
        y = K.conv3x3(x, pk.w, pk.cout, bias=pk.bias, out_f32=True)
        ctx.host = host
        ctx.in_shape = x.shape
        ctx.train_w, ctx.train_b = hw is not None, hb is not None
        if hw is not None:
            ctx.save_for_backward(x)
        return K.nhwc_to_nchw_f32(y, pk.cout)

    @staticmethod
    def backward(ctx, dy):
        pk = ctx.host.packed()
        B, H, W, C = ctx.in_shape
        g = K.nchw_to_nhwc(dy.contiguous(), pk.cout_pad)
        _conv_param_grads(ctx, g, pk)
        dx = K.conv3x3(g, pk.wd, C, mode=1, stride=1, out_hw=(H, W)) if ctx.needs_input_grad[0] else None
        return dx, None, None, None


def _conv_param_grads(ctx, g, pk):
    """Weight / bias gradients of a 3x3 conv whose output gradient `g` is channel-padded NHWC bf16."""
    if ctx.train_w:
        (x,) = ctx.saved_tensors
        K.wgrad_conv3x3(g, x, grad_buffer(ctx.host.weight, True), cout=pk.cout)
    if ctx.train_b:
        gb = grad_buffer(ctx.host.bias)
        if g.shape[-1] == pk.cout:
            K.colsum(g.view(-1, pk.cout), gb)
        else:                                          # conv_out: 4 real channels staged in 8
            tmp = torch.zeros(g.shape[-1], dtype=torch.float32, device=g.device)
            K.colsum(g.view(-1, g.shape[-1]), tmp)
            gb += tmp[:pk.cout]


def conv_out(x, host):
    return _ConvOutFn.apply(x, host, _tr(host.weight), _tr(host.bias))


class _ConvInFn(torch.autograd.Function):
    """conv_in with trainable parameters: the latents carry no gradient, so the parameters are the node's only
    differentiable inputs."""

    @staticmethod
    def forward(ctx, x, host, hw, hb):
        pk = host.packed()
        ctx.host = host
        ctx.train_w, ctx.train_b = hw is not None, hb is not None
        if hw is not None:
            ctx.save_for_backward(x)
        return K.conv3x3(x, pk.w, pk.cout, bias=pk.bias)

    @staticmethod
    def backward(ctx, dy):
        _conv_param_grads(ctx, dy.contiguous(), ctx.host.packed())
        return None, None, None, None


def conv_in(sample, host):
    """NCHW fp32/bf16 latents -> channel-padded NHWC bf16 -> 3x3 conv.  The latents never need a gradient."""
    if sample.requires_grad:
        raise NotImplementedError("hcp_diffusion_amd: gradient w.r.t. the input latents is not implemented")
    pk = host.packed()
    x = K.nchw_to_nhwc(sample.contiguous(), pk.cin_pad)
    hw, hb = _tr(host.weight), _tr(host.bias)
    if hw is None and hb is None:
        return K.conv3x3(x, pk.w, pk.cout, bias=pk.bias)
    return _ConvInFn.apply(x, host, hw, hb)
