"""GPU diagnostic: the reference-style loop over the graphed native UNet (bench.py --seam --seam-graph) with finite checks every step —
some runs of that loop (and one profiled NativeTrainer run) ended with loss = NaN in rounds 3-5.  Reports the first step at which the
prediction, the LoRA gradients or the parameters stop being finite, and what the captured buffers looked like.
   python tools/diag/nan_hunt.py [steps] [graph|eager|trainer]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.optim import FusedAdamW
from hcp_diffusion_amd.scheduler import NativeDDPMScheduler
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
dev = torch.device("cuda:0")
if os.environ.get("HCP_POISON") == "1":
    # NaN-poisoned allocator: fill a large part of HBM with 0xFF bytes (NaN as fp32 and as bf16, -1 as int), then hand the blocks back to
    # torch's caching allocator — every later torch.empty() is carved out of them, so a kernel that reads memory nobody wrote (and
    # multiplies it by zero, or masks it late) turns the step NaN at once instead of once in a while.
    blocks = [torch.full((n,), -1, dtype=torch.int32, device=dev) for n in [2 ** 30] * 24 + [2 ** 26] * 64 + [2 ** 22] * 256 + [2 ** 18] * 512]
    torch.cuda.synchronize()
    del blocks
    print("poisoned", torch.cuda.memory_reserved() / 2 ** 30, "GiB of cached blocks", flush=True)
torch.manual_seed(114514)
with torch.device("meta"):
    unet = NativeUNet2DConditionModel()
unet = unet.to_empty(device=dev)
with torch.no_grad():
    for name, p in unet.named_parameters():
        if p.dim() > 1:
            p.normal_(0, p[0].numel() ** -0.5)
        elif "norm" in name and name.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
tr = NativeTrainer(unet, [dict(layers=PATS, rank=8, lr=1e-4)], lr=1e-4, weight_decay=1e-3, scale_lr_factor=4, use_graph=(mode == "trainer"))
torch.manual_seed(114514)
with torch.no_grad():
    for blk in tr.bucket.blocks:
        blk.layer.W_up.normal_(0, 0.02)
tr.bucket.pack()
B = 4
latents = torch.randn(B, 4, 64, 64, device=dev)
ehs = torch.randn(B, 77, 768, device=dev).to(torch.bfloat16)
fin = lambda t: bool(torch.isfinite(t).all().item())
if mode == "trainer":
    for i in range(steps):
        loss = tr.train_one_step(latents, ehs)
        torch.cuda.synchronize()
        ok = (fin(loss), fin(tr.bucket.params), fin(tr.exp_avg))
        print(f"step {i}: loss {loss.item():.5f} finite(loss, params, exp_avg) {ok}", flush=True)
        if not all(ok):
            break
    sys.exit(0)
sched = NativeDDPMScheduler()
params = [p for blk in tr.bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
opt = FusedAdamW([dict(params=params, lr=1e-4 * B)], weight_decay=1e-3)
crit = torch.nn.MSELoss(reduction="none")
if mode == "graph":
    unet.enable_hip_graph()
first_bad = []
if os.environ.get("HCP_HOOKS") == "1":              # eager only: name the first module whose output / input-gradient is not finite
    def fwd_hook(name):
        def f(m, a, out):
            y = out[0] if isinstance(out, tuple) else (out.sample if hasattr(out, "sample") else out)
            if torch.is_tensor(y) and y.is_floating_point() and not first_bad and not fin(y):
                first_bad.append(("forward", name, type(m).__name__)); print("  FIRST non-finite forward output:", name, type(m).__name__, flush=True)
        return f
    def bwd_hook(name):
        def f(m, gin, gout):
            for g in gin:
                if torch.is_tensor(g) and not fin(g) and not first_bad:
                    first_bad.append(("backward", name, type(m).__name__)); print("  FIRST non-finite input gradient:", name, type(m).__name__, flush=True)
        return f
    for n, m in unet.named_modules():
        if n and not any(True for _ in m.children()):
            m.register_forward_hook(fwd_hook(n)); m.register_full_backward_hook(bwd_hook(n))
for i in range(steps):
    noise = torch.randn_like(latents)
    t = torch.randint(0, 1000, (B,), device=dev).long()
    pred = unet(sched.add_noise(latents, noise, t), t, ehs).sample
    loss = crit(pred.float(), noise.float()).mean()
    loss.backward()
    torch.cuda.synchronize()
    g_ok = fin(tr.bucket.grads); gn = float(tr.bucket.grads.norm())
    bad_blocks = []
    if not g_ok:
        for blk in tr.bucket.blocks:
            for nm, p in (("W_down", blk.layer.W_down), ("W_up", blk.layer.W_up)):
                if p.grad is not None and not fin(p.grad):
                    bad_blocks.append((blk.name if hasattr(blk, "name") else "?", nm, int((~torch.isfinite(p.grad)).sum())))
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    print(f"step {i}: t {t.tolist()} loss {loss.item():.5f} finite pred {fin(pred)} grads {g_ok} (|g| {gn:.4e}) params {fin(tr.bucket.params)}", flush=True)
    if bad_blocks:
        names = {id(b): n for n, b in tr.lora_group.plugin_dict.items()} if hasattr(tr, "lora_group") else {}
        print("  non-finite gradient tensors:", len(bad_blocks), bad_blocks[:8], flush=True)
    if not (fin(pred) and g_ok and fin(tr.bucket.params)):
        break
