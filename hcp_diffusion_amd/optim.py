"""Seam 4 (``train.optimizer``, train_base.yaml:37-40 -> train_ac.py:370: ``cfg.train.optimizer(params=params_group)``): a
``torch.optim.Optimizer`` whose step is the fused AdamW kernel (csrc/optim.hip).

The reference trainer clips with ``accelerator.clip_grad_norm_`` before ``optimizer.step()`` (train_ac.py:485-491), so the step
here is plain AdamW (decoupled weight decay, bias correction as torch.optim.AdamW).  Parameters whose storage is adjacent — all
``W_down`` / ``W_up`` of a native LoRA model live in ONE flat bucket, a fully fine-tuned UNet's in another — are stepped with one
launch per contiguous run per param group instead of one per tensor (320 launches -> 1 for the conventional LoRA config).
The kernel leaves the consumed gradients zeroed (the reference's ``zero_grad`` that follows is then a no-op on values)."""
import torch

from . import kernels as K


def _dense_flat(t):
    """1-D view over the dense storage range of a contiguous / channels_last tensor (element order is irrelevant to AdamW)."""
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        raise ValueError("FusedAdamW: parameters and gradients must be dense (contiguous or channels_last)")
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._runs = {}            # group index -> (sampled signature, [run], full signature)
        self._nsteps = {}          # group index -> steps since the runs were built (full pointer check every FULL_CHECK_EVERY)

    def _build_runs(self, group):
        items = []
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise TypeError("FusedAdamW keeps fp32 master parameters (the reference: fp32 params under autocast)")
            items.append((p.untyped_storage().data_ptr(), p.storage_offset(), p.numel(), p,
                          p.grad.untyped_storage().data_ptr(), p.grad.storage_offset()))
        items.sort(key=lambda it: (it[0], it[1]))
        runs, cur = [], None
        for sp, so, n, p, gp, go in items:
            if cur and cur["sp"] == sp and cur["gp"] == gp and cur["so"] + cur["n"] == so and cur["go"] + cur["n"] == go:
                cur["n"] += n
            else:
                cur = dict(sp=sp, gp=gp, so=so, go=go, n=n, p=p)
                runs.append(cur)
        out = []
        for r in runs:
            p, n = r["p"], r["n"]
            pf = _dense_flat(p).as_strided((n,), (1,), r["so"])
            gf = _dense_flat(p.grad).as_strided((n,), (1,), r["go"])
            out.append(dict(p=pf, g=gf, m=torch.zeros_like(pf), v=torch.zeros_like(pf),
                            lr=torch.zeros(1, dtype=torch.float32, device=pf.device), step=torch.zeros(1, dtype=torch.int32, device=pf.device)))
        return out

    FULL_CHECK_EVERY = 64

    @staticmethod
    def _sig(group, full=False):
        """What the flat runs were built from: which parameters have gradients and where both live.  Per step only a sample is compared
        (count + first / middle / last pointers: a re-homed bucket or a dropped gradient moves those); 640 data_ptr() calls per step
        were 0.25 ms of the reference-style loop's host time.  full=True compares every pointer: step() does that every
        FULL_CHECK_EVERY steps, so an INTERIOR gradient re-allocated outside the flat run (p.grad = None on one layer, a module
        swapping a tensor) while the count stays the same is still caught instead of stepping from a stale slice for the whole run."""
        ps = [p for p in group["params"] if p.grad is not None]
        pick = ps if full else ([ps[0], ps[len(ps) // 2], ps[-1]] if ps else [])
        return (len(ps),) + tuple((p.data_ptr(), p.grad.data_ptr()) for p in pick)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            sig = self._sig(group)
            cached = self._runs.get(gi)
            moved = cached is not None and cached[0] != sig
            if cached is not None and not moved:
                self._nsteps[gi] = self._nsteps.get(gi, 0) + 1
                if self._nsteps[gi] % self.FULL_CHECK_EVERY == 0:
                    moved = self._sig(group, full=True) != cached[2]
            if cached is None or moved:
                if cached is not None:
                    raise RuntimeError("FusedAdamW: the set of parameters with gradients (or their storage) changed between steps; "
                                       "the flat AdamW moments cannot follow")
                cached = (sig, self._build_runs(group), self._sig(group, full=True))
                self._runs[gi] = cached
                self._apply_resume(group, cached[1])
            b1, b2 = group["betas"]
            for r in cached[1]:
                r["lr"].fill_(group["lr"])
                K.adamw_clip_fused(r["p"], r["g"], r["m"], r["v"], r["lr"], r["step"], beta1=b1, beta2=b2, eps=group["eps"],
                                   weight_decay=group["weight_decay"], sumsq_t=None, grad_scale=1.0, max_norm=0.0)
        self._bump_versions()
        return loss

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad walks the parameters one by one (1.0 ms of host time per step for the 320 LoRA tensors under the reference's
        loop, train_ac.py:494); the gradients of a run are ONE flat range, so `set_to_none=False` is one fill per run.  (After step() the
        fused kernel has already left them zero; a backward that ran in between is why this still writes.)  set_to_none=True keeps torch's
        behaviour: the native layers / graphed modules re-attach and clear the dropped views themselves."""
        if set_to_none:
            return super().zero_grad(set_to_none=True)
        covered = set()
        for gi, group in enumerate(self.param_groups):
            cached = self._runs.get(gi)
            if cached is None:
                continue
            if self._sig(group) != cached[0]:
                continue                                     # storage moved since the runs were built: let torch do it
            for r in cached[1]:
                r["g"].zero_()
            covered.update(id(p) for p in group["params"] if p.grad is not None)
        rest = [p for group in self.param_groups for p in group["params"] if p.grad is not None and id(p) not in covered]
        for p in rest:
            p.grad.detach_(); p.grad.requires_grad_(False); p.grad.zero_()

    def _bump_versions(self):
        """The kernel wrote through raw pointers: advance the parameters' autograd version counters, as the in-place torch ops of
        torch.optim.AdamW would have (the native layers key their bf16 operand caches on them).  No kernel, no data touched."""
        ps = [p for group in self.param_groups for p in group["params"] if p.grad is not None]
        fn = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
        if fn is not None:
            try:
                fn(ps, [p._version + 1 for p in ps])
                return
            except TypeError:                              # earlier 2.x: (Tensor, int) per call
                try:
                    for p in ps:
                        fn(p, p._version + 1)
                    return
                except TypeError:
                    pass
        for p in ps:                                       # any torch: an in-place op on an empty slice bumps the counter
            p[:0].add_(0)

    # ---- checkpoint / resume: the flat moments live outside `self.state` (one tensor per contiguous run, not per parameter)
    def state_dict(self):
        """torch's layout ({'state': {index: {...}}, 'param_groups': [...]}) with PER-PARAMETER exp_avg / exp_avg_sq / step cut out of the
        flat runs, so that the file resumes a torch.optim.AdamW as well as a FusedAdamW (train_ac.py:370 builds either from the same cfg)."""
        base = super().state_dict()
        state, index = {}, 0
        ids = {}
        for group in self.param_groups:
            for p in group["params"]:
                ids[id(p)] = index; index += 1
        for gi, group in enumerate(self.param_groups):
            cached = self._runs.get(gi)
            if cached is None:
                continue
            for r in cached[1]:
                base_ptr, step = r["p"].data_ptr(), r["step"].item()
                for p in group["params"]:
                    off = (p.data_ptr() - base_ptr) // 4
                    if p.grad is None or off < 0 or off + p.numel() > r["p"].numel() or p.untyped_storage().data_ptr() != r["p"].untyped_storage().data_ptr():
                        continue
                    view = lambda flat: flat[off:off + p.numel()].as_strided(p.shape, p.stride()).clone()
                    state[ids[id(p)]] = {"step": torch.tensor(float(step)), "exp_avg": view(r["m"]), "exp_avg_sq": view(r["v"])}
        base["state"] = state
        return base

    def load_state_dict(self, state_dict):
        """Moments and step counts of a file written by state_dict() above or by torch.optim.AdamW over the same parameters; the flat
        runs are (re)built on the first step, so the values wait in `_resume` until then."""
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        order = [p for group in self.param_groups for p in group["params"]]
        self._resume = {id(order[int(i)]): st for i, st in state_dict.get("state", {}).items()}
        self._runs = {}

    def _apply_resume(self, group, runs):
        res = getattr(self, "_resume", None)
        if not res:
            return
        for r in runs:
            base_ptr = r["p"].data_ptr()
            for p in group["params"]:
                st = res.get(id(p))
                off = (p.data_ptr() - base_ptr) // 4
                if st is None or p.grad is None or off < 0 or off + p.numel() > r["p"].numel() or \
                        p.untyped_storage().data_ptr() != r["p"].untyped_storage().data_ptr():
                    continue
                for name, flat in (("exp_avg", r["m"]), ("exp_avg_sq", r["v"])):
                    flat[off:off + p.numel()].as_strided(p.shape, p.stride()).copy_(st[name].to(flat.device))
                r["step"].fill_(int(float(st["step"])))
