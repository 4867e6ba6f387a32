"""hipGraph replay of the UNet's forward and backward for trainers that call the module the ordinary way — the reference's
`Trainer.train_one_step` (train_ac.py:467-504: `pred = TE_unet(...)`, `loss.backward()`, clip, `optimizer.step()`): eager, the ~1000
kernel launches of a step cost 40-49 ms of Python + launch path against 20 ms of GPU time.

`unet.enable_hip_graph()` makes `unet(sample, t, ehs, ...)` (grad mode, GPU tensors) run as TWO captured graphs per input signature:
forward at the call, backward when autograd reaches the node (`torch.cuda.make_graphed_callables` cannot be used: the native layers
write the LoRA gradients in place into the flat bucket instead of returning them to autograd).  Everything else — loss, clipping,
optimizer, scheduler, checkpointing — stays the trainer's.  Falls back to the eager path, silently and per call, when the call is
not capturable: no grad mode, host (non-LoRA) parameters training, LoRA dropout active, forward hooks on the UNet (ControlNet feeder).
"""
import torch

from . import kernels as K  # noqa: F401  (the library must be loaded before any capture)


class _Entry:
    __slots__ = ("g_fwd", "g_bwd", "static_in", "out", "dout", "grad_in", "buckets", "blocks", "keep", "pending")


def _lora_buckets(unet):
    from .lora import LoraHipLayer
    blocks = [m for m in unet.modules() if isinstance(m, LoraHipLayer)]
    buckets = []
    for b in blocks:
        bk = getattr(b, "_bucket", None)
        if bk is not None and all(bk is not x for x in buckets):
            buckets.append(bk)
    return blocks, buckets


def capturable(unet):  # (any native module that holds LoRA layers: the UNet or the text encoder)
    """LoRA-only training (frozen host), no active dropout, no hooks feeding per-step data."""
    from .lora import LoraBucket, LoraHipLayer
    lora_params = set()
    loose = [m for m in unet.modules() if isinstance(m, LoraHipLayer) and m._bucket is None]
    if loose:                                       # blocks built by the reference's make_hcpdiff: one flat bucket for all of them
        LoraBucket(loose)                           # (the Parameters keep their identity: an optimizer that already holds them is unaffected)
    for m in unet.modules():
        if isinstance(m, LoraHipLayer):
            lora_params.update(id(p) for p in m.parameters())
        if isinstance(m, torch.nn.Dropout) and m.p > 0 and m.training:
            return False
    if unet._forward_pre_hooks or unet._forward_hooks:
        return False
    d = torch.distributed
    if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
        return False                                # under torch DDP the reducer's per-parameter hooks must run with every backward: eager module
    any_lora = False
    for p in unet.parameters():
        if p.requires_grad:
            if id(p) not in lora_params:
                return False
            any_lora = True
    return any_lora


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, anchor, *inputs):      # anchor: any trainable leaf, so that autograd schedules backward()
        for s, t in zip(entry.static_in, inputs):
            if s is not None:
                s.copy_(t)
        for bk in entry.buckets:                   # an optimizer stepped the fp32 factors: refresh the bf16 operands (one launch)
            if bk._stale(bk.blocks):
                bk.pack()
        entry.g_fwd.replay()
        entry.pending = True                        # the graph's saved activations now belong to THIS call until its backward ran
        ctx.entry = entry
        return entry.out.detach().clone()          # the caller may keep the prediction across the next replay

    @staticmethod
    def backward(ctx, dy):
        e = ctx.entry
        e.dout.copy_(dy)
        e.g_bwd.replay()
        e.pending = False
        for bk in e.buckets:                       # zero_grad(set_to_none=True) drops the .grad views; the kernels wrote into the bucket
            for b in bk.blocks:
                bk.grad_views_for(b)
        grads = [None, None]
        for s, g in zip(e.static_in, e.grad_in):
            grads.append(g.clone() if g is not None else None)
        return tuple(grads)


def capture(unet, inputs, fwd):
    """inputs: tensors or None in the order of `fwd`'s positional arguments; fwd(*static) -> prediction tensor."""
    dev = next(t for t in inputs if t is not None).device
    e = _Entry()
    e.blocks, e.buckets = _lora_buckets(unet)
    e.static_in = [None if t is None else t.detach().clone().requires_grad_(t.requires_grad and t.is_floating_point()) for t in inputs]
    saved = [(p, p.grad.detach().clone()) for p in unet.parameters() if p.requires_grad and p.grad is not None]
    bucket_grads = [bk.grads.detach().clone() for bk in e.buckets if hasattr(bk, "grads")]
    from . import ops
    grouped_was = ops._group["enabled"]
    ops.enable_grouped_wgrad(True)                  # all LoRA weight gradients of the backward as ONE launch at its end (as NativeTrainer does)

    def backward(out, dout):
        K.wgrad_staging_begin_step()                # descriptor staging: the capture must find its slots allocated by the warm-up
        torch.autograd.backward(out, dout)
        ops.flush_grouped_wgrad()

    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    try:
        with torch.cuda.stream(s):
            for _ in range(2):                      # warm-up: lazy packing, workspaces, caches — nothing may allocate / pack under capture
                out = fwd(*e.static_in)
                backward(out, torch.zeros_like(out))
                for t in e.static_in:
                    if t is not None and t.requires_grad:
                        t.grad = None
            e.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g_fwd, stream=s):
                e.out = fwd(*e.static_in)
            e.dout = torch.zeros_like(e.out)
            e.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g_bwd, pool=e.g_fwd.pool(), stream=s):
                backward(e.out, e.dout)
            e.keep = ops._group["keep"]             # descriptor table + operand tensors of the captured grouped launch
            e.grad_in = [None if (t is None or not t.requires_grad) else t.grad for t in e.static_in]
    finally:
        ops.enable_grouped_wgrad(grouped_was)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    for p, g in saved:                              # the warm-up and the capture accumulated into the gradient buckets: undo
        if p.grad is None:
            p.grad = g
        else:
            p.grad.copy_(g)
    for bk, g in zip([bk for bk in e.buckets if hasattr(bk, "grads")], bucket_grads):
        bk.grads.copy_(g)
    return e


def call(unet, inputs, fwd, cache, key):
    e = cache.get(key)
    if e is None:
        e = cache[key] = capture(unet, inputs, fwd)
        e.pending = False
    if e.pending:                                   # a second forward before the first one's backward (two losses summed, then one
        return fwd(*inputs)                         # backward): the graph holds ONE set of activations, so this call runs eagerly
    anchor = next(p for b in e.blocks for p in b.parameters() if p.requires_grad)
    return _GraphedFn.apply(e, anchor, *[t for t in inputs])
