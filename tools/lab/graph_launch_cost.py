"""GPU lab: how long does the HOST spend inside hipGraphLaunch for the step's graphs, and how long after the call does the GPU start?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from hcp_diffusion_amd.trainer import NativeTrainer
dev = torch.device("cuda:0")
with torch.device("meta"):
    unet = NativeUNet2DConditionModel()
unet = unet.to_empty(device=dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for p in unet.parameters():
        p.normal_(0, 0.02, generator=g)
pats = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
tr = NativeTrainer(unet, [dict(layers=pats, rank=8)], lr=1e-4, use_graph=True)
lat = torch.randn(4, 4, 64, 64, device=dev); ehs = torch.randn(4, 77, 768, device=dev).to(torch.bfloat16)
for _ in range(3):
    tr.train_one_step(lat, ehs)
torch.cuda.synchronize()
graph = next(iter(tr._graph_cache.values()))[0]
for name, gr in (("forward+backward graph", graph), ("optimizer graph", tr._opt_graph)):
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); gr.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    print(f"{name}: host inside replay() {sorted(host)[5]:.3f} ms, launch -> done {sorted(total)[5]:.3f} ms")
# module-level graphs (the seam path)
unet.enable_hip_graph()
x = torch.randn(4, 4, 64, 64, device=dev); t = torch.randint(0, 1000, (4,), device=dev)
for _ in range(3):
    unet(x, t, ehs).sample.float().square().mean().backward()
torch.cuda.synchronize()
e = next(iter(unet._hip_graphs.values()))
for name, gr in (("module forward graph", e.g_fwd), ("module backward graph", e.g_bwd)):
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); gr.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    print(f"{name}: host inside replay() {sorted(host)[5]:.3f} ms, launch -> done {sorted(total)[5]:.3f} ms")

# ---- the reference-style loop over the graphed module: where do the milliseconds above the two graphs go?
from hcp_diffusion_amd.optim import FusedAdamW
from hcp_diffusion_amd.scheduler import NativeDDPMScheduler
sched = NativeDDPMScheduler()
params = [p for blk in tr.bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
opt = FusedAdamW([dict(params=params, lr=1e-4)], weight_decay=1e-3)
crit = torch.nn.MSELoss(reduction="none")
acc = {}
def phase(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3; return r
def step(timed):
    ph = phase if timed else (lambda n, f: f())
    noise, tt = ph("noise draw", lambda: (torch.randn_like(x), torch.randint(0, 1000, (4,), device=dev).long()))
    noisy = ph("add_noise", lambda: sched.add_noise(x, noise, tt))
    pred = ph("unet forward (graph)", lambda: unet(noisy, tt, ehs).sample)
    loss = ph("loss", lambda: crit(pred.float(), noise.float()).mean())
    ph("backward (graph)", lambda: loss.backward())
    ph("clip_grad_norm_", lambda: torch.nn.utils.clip_grad_norm_(params, 1.0))
    ph("optimizer.step", lambda: opt.step())
    ph("zero_grad", lambda: opt.zero_grad(set_to_none=False))
    return ph("loss.item", lambda: loss.item())
for _ in range(5):
    step(False)
N = 20
for _ in range(N):
    step(True)
print("reference-style loop, per phase (sync after each), ms/step:")
for k, v in acc.items():
    print(f"  {k:24s} {v / N:7.3f}")
print(f"  {'sum':24s} {sum(acc.values()) / N:7.3f}")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    step(False)
torch.cuda.synchronize(); print(f"  un-instrumented loop      {(time.perf_counter() - t0) / N * 1e3:7.3f}")
