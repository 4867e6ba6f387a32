"""hipGraph replay of a native module's forward and backward for trainers that call the module the ordinary way — the reference's
`Trainer.train_one_step` (train_ac.py:467-504: `pred = TE_unet(...)`, `loss.backward()`, clip, `optimizer.step()`), alone or under
`accelerate launch` with torch DDP around the model (train_ac.py:116-123,175): eager, the ~1000 kernel launches of a step cost 40-49 ms
of Python + launch path against 20 ms of GPU time.

`unet.enable_hip_graph()` makes `unet(sample, t, ehs, ...)` (grad mode, GPU tensors) run as TWO captured graphs per input signature:
forward at the call, backward when autograd reaches the node (`torch.cuda.make_graphed_callables` cannot be used: the native layers
write parameter gradients in place into flat buckets instead of returning them to autograd).  Everything else — loss, clipping,
optimizer, scheduler, gradient exchange — stays the trainer's.

What trains may be LoRA blocks (their flat `LoraBucket`), host parameters (full fine-tune, DreamBooth.yaml:6-10: re-homed into one flat
`fullft.HostBucket` the first time the module is captured; the Parameters keep their identity, so an optimizer that already holds them
is unaffected), or both.  Every trainable parameter is an INPUT of the autograd node: the engine then runs each parameter's
AccumulateGrad node after the backward graph was enqueued (with an undefined gradient — the kernels already wrote `.grad`), which is
what torch DDP's reducer hooks hang on; stock DDP therefore averages the bucket views exactly as it does for the eager module
(tests/test_dist.py).  Falls back to the eager path, per call, when the call is not capturable: no grad mode, LoRA dropout active,
forward hooks on the module (ControlNet's feeder / branch hooks run per-step Python).

One memory pool is shared by all signatures of a module (one forward/backward pair runs at a time), and at most `MAX_SIGNATURES`
pairs are kept (least recently used goes first): the reference's aspect-ratio buckets hand a run tens of resolutions.
"""
import weakref

import torch

from . import kernels as K  # noqa: F401  (the library must be loaded before any capture)

MAX_SIGNATURES = 8


class _Entry:
    __slots__ = ("g_fwd", "g_bwd", "static_in", "out", "dout", "grad_in", "buckets", "blocks", "host", "params", "keep", "pending",
                 "pending_node", "live", "wg", "fwd", "__weakref__")


class _Recorded:
    """CPU stand-in for a captured graph (tests: interpreter backend, gloo): `replay()` re-runs the recorded callable on the static
    tensors.  Same bookkeeping, same autograd wiring, no hipGraph."""

    def __init__(self, fn):
        self.fn = fn

    def replay(self):
        self.fn()

    def pool(self):
        return None


def _lora_buckets(unet):
    from .lora import LoraHipLayer
    blocks = [m for m in unet.modules() if isinstance(m, LoraHipLayer)]
    buckets = []
    for b in blocks:
        bk = getattr(b, "_bucket", None)
        if bk is not None and all(bk is not x for x in buckets):
            buckets.append(bk)
    return blocks, buckets


def _host_bucket(unet, host_params):
    """The flat bucket of the trainable HOST parameters (built once per set of parameters; NativeTrainer's own bucket is reused)."""
    from .fullft import HostBucket
    key = tuple(id(p) for _, p in host_params)
    hit = getattr(unet, "_hcp_host_bucket", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    owner = {id(p): getattr(p, "_hcp_bucket", None) for _, p in host_params}
    owners = {id(o) for o in owner.values()}
    if len(owners) == 1 and None not in owner.values():          # already flat (a NativeTrainer built it)
        hb = next(iter(owner.values()))
    else:
        hb = HostBucket(unet, list(host_params))
    unet._hcp_host_bucket = (key, hb)
    return hb


def capturable(unet):  # (any native module that holds LoRA layers and / or trainable host parameters: the UNet or the text encoder)
    """No active dropout, no hooks feeding per-step data, something to train."""
    from .lora import LoraBucket, LoraHipLayer
    loose = [m for m in unet.modules() if isinstance(m, LoraHipLayer) and m._bucket is None]
    if loose:                                       # blocks built by the reference's make_hcpdiff: one flat bucket for all of them
        LoraBucket(loose)                           # (the Parameters keep their identity: an optimizer that already holds them is unaffected)
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout) and m.p > 0 and m.training:
            return False
    if unet._forward_pre_hooks or unet._forward_hooks:
        return False
    lora_ids = {id(p) for m in unet.modules() if isinstance(m, LoraHipLayer) for p in m.parameters()}
    trainable = [p for p in unet.parameters() if p.requires_grad]
    if any(p.dtype != torch.float32 for p in trainable if id(p) not in lora_ids):
        return False                                # HostBucket keeps fp32 masters only: a non-fp32 trainable host parameter stays eager
    return bool(trainable)


def capturable_cached(mod):
    """(capturable(mod), any submodule checkpointing) without walking the module tree on every call (3 walks over ~1000 modules cost
    4.4 ms per forward under the reference-style loop: tools/lab/graph_launch_cost.py).  Re-evaluated when the cheap tell-tales move —
    train()/eval() (LoRA dropout), hooks attached to / removed from the module (ControlNet feeders) — and by `enable_hip_graph()` /
    `reset_hip_graph()` / `enable_gradient_checkpointing()`, which the caller runs after changing what trains (unet.py docstrings)."""
    # (requires_grad_() flips after the first call are NOT seen here — a walk over the parameters is the cost this cache removes:
    #  call reset_hip_graph() after changing what trains)
    key = (mod.training, len(mod._forward_pre_hooks), len(mod._forward_hooks))
    hit = getattr(mod, "_hcp_capturable", None)
    if hit is None or hit[0] != key:
        hit = (key, capturable(mod), any(getattr(m, "gradient_checkpointing", False) for m in mod.modules()))
        mod._hcp_capturable = hit
    return hit[1], hit[2]


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, n_params, *args):      # args = the trainable parameters (autograd schedules backward; DDP's hooks fire), then the inputs
        inputs = args[n_params:]
        for s, t in zip(entry.static_in, inputs):
            if s is not None:
                s.copy_(t)
        for bk in entry.buckets:                   # an optimizer stepped the fp32 factors: refresh the bf16 operands (one launch)
            if bk._stale(bk.blocks):
                bk.pack()
        if entry.host is not None and entry.host.stale():
            entry.host.repack()                    # ... and the bf16 host operands (one grouped launch)
        entry.g_fwd.replay()
        entry.pending = True                        # the graph's saved activations now belong to THIS call until its backward ran
        ctx.entry = entry
        return entry.out.detach().clone()          # the caller may keep the prediction across the next replay

    @staticmethod
    def backward(ctx, dy):
        e = ctx.entry
        e.dout.copy_(dy)
        # zero_grad(set_to_none=True) (torch's default) DROPS the .grad views: the trainer's "zero" never reached the buckets the
        # captured kernels accumulate into.  Clear what was dropped, then replay, then hand the views back.
        for bk in e.buckets:
            bk.zero_dropped()
        if e.host is not None:
            e.host.zero_dropped()
        e.g_bwd.replay()
        e.pending = False
        for bk in e.buckets:
            for b in bk.blocks:
                bk.grad_views_for(b)
        if e.host is not None:
            e.host.attach_grads()
        grads = [None, None] + [None] * len(e.params)
        for s, g in zip(e.static_in, e.grad_in):
            grads.append(g.clone() if g is not None else None)
        return tuple(grads)


def capture(unet, inputs, fwd, pool=None):
    """inputs: tensors or None in the order of `fwd`'s positional arguments; fwd(*static) -> prediction tensor."""
    from . import ops
    dev = next(t for t in inputs if t is not None).device
    on_gpu = dev.type == "cuda"
    e = _Entry()
    e.blocks, e.buckets = _lora_buckets(unet)
    lora_ids = {id(p) for b in e.blocks for p in b.parameters()}
    host_params = [(n, p) for n, p in unet.named_parameters() if p.requires_grad and id(p) not in lora_ids]
    e.host = _host_bucket(unet, host_params) if host_params else None
    e.params = [p for p in unet.parameters() if p.requires_grad]
    e.static_in = [None if t is None else t.detach().clone().requires_grad_(t.requires_grad and t.is_floating_point()) for t in inputs]
    bucket_grads = [bk.grads.detach().clone() for bk in e.buckets] + ([e.host.grads.detach().clone()] if e.host is not None else [])
    e.wg = ops.WgradContext(grouped=True)           # all LoRA weight gradients of the backward as ONE launch at its end (as NativeTrainer does)
    e.live = None

    def run_fwd():
        with ops.wgrad_context(e.wg), torch.enable_grad():
            return fwd(*e.static_in)

    def run_bwd(out, dout):
        K.wgrad_staging_begin_step()                # descriptor staging: the capture must find its slots allocated by the warm-up
        torch.autograd.backward(out, dout)
        e.wg.flush()

    def clear_input_grads():
        for t in e.static_in:
            if t is not None and t.requires_grad:
                t.grad = None

    if on_gpu:
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):                      # warm-up: lazy packing, workspaces, caches — nothing may allocate / pack under capture
                out = run_fwd()
                run_bwd(out, torch.zeros_like(out))
                clear_input_grads()
            e.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g_fwd, stream=s, pool=pool):
                e.out = run_fwd()
            e.dout = torch.zeros_like(e.out)
            e.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g_bwd, pool=e.g_fwd.pool(), stream=s):
                run_bwd(e.out, e.dout)
            e.grad_in = [None if (t is None or not t.requires_grad) else t.grad for t in e.static_in]
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
    else:                                           # interpreter backend: recorded callables instead of graphs
        # A replayed hipGraph involves no autograd; the recorded stand-in re-runs the eager module INSIDE the outer backward, and the
        # real Parameters' AccumulateGrad nodes (DDP's hooks hang on them) must not run a second time there.  The recorded forward
        # therefore sees storage-sharing ALIASES of the trainable parameters (their .grad = the same bucket views).
        from torch.nn.utils.stateless import _reparametrize_module
        names = {id(p): n for n, p in unet.named_parameters()}
        gviews = {id(p): g for p, g in e.host._gv} if e.host is not None else {}
        plain_fwd = run_fwd

        def run_fwd():                              # noqa: F811
            aliases = {}
            for p in e.params:
                a = p.detach().requires_grad_(True)
                if id(p) in gviews:
                    a.grad = gviews[id(p)]
                aliases[names[id(p)]] = a
            with _reparametrize_module(unet, aliases):
                return plain_fwd()
        out = run_fwd()
        run_bwd(out, torch.zeros_like(out))
        e.out = out.detach().clone()
        e.dout = torch.zeros_like(e.out)
        e.grad_in = [None if (t is None or not t.requires_grad) else torch.zeros_like(t) for t in e.static_in]
        clear_input_grads()

        def rec_fwd():
            e.live = run_fwd()
            e.out.copy_(e.live.detach())

        def rec_bwd():
            clear_input_grads()
            run_bwd(e.live, e.dout)
            for t, g in zip(e.static_in, e.grad_in):
                if g is not None:
                    g.copy_(t.grad)
            e.live = None
        e.g_fwd, e.g_bwd = _Recorded(rec_fwd), _Recorded(rec_bwd)
    e.keep = e.wg.keep                              # descriptor table + operand tensors of the captured grouped launch
    # the warm-up and the capture accumulated into the gradient buckets: undo
    for bk, g in zip(list(e.buckets) + ([e.host] if e.host is not None else []), bucket_grads):
        bk.grads.copy_(g)
    e.pending, e.pending_node = False, None
    return e


_warned = set()


def set_max_signatures(mod, n):
    """Overlay key `hip_graph_max_signatures` (next to `hip_graph`; `enable_hip_graph(max_signatures=n)`): how many input signatures
    (aspect-ratio buckets x datasets, data/bucket.py:63-229) of THIS module keep their captured pair; None = `MAX_SIGNATURES`.  Past it
    the least recently used pair is dropped and re-captured on its next use (2 warm-up steps + 2 captures), which `call` reports once
    per module."""
    if n is not None and int(n) < 1:
        raise ValueError("hip_graph_max_signatures must be >= 1")
    mod._hip_graph_max = None if n is None else int(n)


def _live_pending(e):
    """True while a forward of this entry still awaits its backward (its autograd node is alive)."""
    if not e.pending:
        return False
    node = e.pending_node() if e.pending_node is not None else None
    if node is None:
        e.pending = False                           # the call's graph was dropped without a backward (eval in grad mode, an exception)
        return False
    return True


def _warn_once(unet, tag, msg):
    if (id(unet), tag) not in _warned:
        _warned.add((id(unet), tag))
        import warnings
        warnings.warn("hcp_diffusion_amd: " + msg)


def call(unet, inputs, fwd, cache, key):
    # All signatures of a module capture into ONE memory pool: the activations a pending forward saved live in pool blocks that any
    # other signature's replay (or a new capture) is free to overwrite.  So while ANY entry awaits its backward, nothing is replayed,
    # captured or evicted: that call runs eagerly (fwd(shape A) -> fwd(shape B) -> backward stays correct).
    if any(_live_pending(x) for x in cache.values()):
        _warn_once(unet, "pending", "a second forward before the pending backward runs eagerly (hip_graph holds one set of activations)")
        return fwd(*inputs)
    e = cache.get(key)
    if e is None:
        pool = next((x.g_fwd.pool() for x in cache.values()), None)       # one pool for every signature of this module
        cap = getattr(unet, "_hip_graph_max", None) or MAX_SIGNATURES
        while len(cache) >= cap:
            cache.pop(next(iter(cache)))                                    # least recently used first (dict order = use order, below)
            _warn_once(unet, "evict", f"more than {cap} input signatures: the least recently used hipGraph pair is dropped and "
                                      "re-captured on its next use (raise `hip_graph_max_signatures`)")
        e = capture(unet, inputs, fwd, pool)
    else:
        cache.pop(key)
    cache[key] = e                                                          # most recently used last
    out = _GraphedFn.apply(e, len(e.params), *e.params, *inputs)
    e.pending_node = weakref.ref(out.grad_fn) if out.grad_fn is not None else None
    return out
