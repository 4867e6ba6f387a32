"""GPU-box diagnostic (WHERE is the native SDXL forward noisier than the reference's own bf16 mode?).

For every ResnetBlock2D, Transformer2DModel and down/up-sampler of the SDXL UNet (configs[3]: B = 2, 128x128 latents, the fixture's inputs)
the fp32 oracle's INPUT of that module (recorded in one fp32 oracle forward on the host cores) is fed to
  (a) the native module (bf16 kernels on the GPU; forward pre-hooks swap the oracle's input in during one native forward), and
  (b) the same oracle module under torch.autocast(bfloat16) — the reference's execution mode (train_ac.py:449),
and both outputs are compared with the fp32 oracle's output of the module: per-module rel-L2 of each and their ratio.  Nothing accumulates:
a ratio well above 1 names the module kind whose native arithmetic is less precise than autocast's.
   python tools/diag/sdxl_block_diag.py [sd15] [lora] [stream-off]
        sd15: the SD1.5 B = 4 benchmark shape instead.
        lora (round 6): both models carry the LoRA of the configuration (rank 16 / 8 on attention + feed-forward, seeded non-zero factors) — the
        oracle through the reference-form layers (oracle/lora_ref.py: mm + fp32 bias, so (b) is the reference's MIXED fp32 / bf16 mode, the
        baseline of tests/test_full_configs.py), the native one through NativeTrainer's LoRA blocks; stream-off: native transformer stacks
        with the plain bf16 residual stream (default: (hi | lo) pair where the model turns it on).
"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K                                     # noqa: E402
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel                  # noqa: E402
from oracle.make_golden import sd15_b4_inputs, sdxl_b2_inputs                   # noqa: E402
from oracle.unet_sd15 import SDXL_CONFIG, OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod, seeded_init_   # noqa: E402

sd15 = "sd15" in sys.argv[1:]
with_lora = "lora" in sys.argv[1:]
smoke = os.environ.get("HCP_DIAG_EMU") == "1"                   # CPU smoke test of this script: interpreter kernels, tiny SDXL config
dev = torch.device("cpu" if smoke else "cuda:0")
cfg = {} if sd15 else SDXL_CONFIG
if smoke:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import emu_cdll
    from oracle.unet_sd15 import TINY_SDXL_CONFIG
    K._set_backend_for_tests(emu_cdll())
    cfg = TINY_SDXL_CONFIG
t0 = time.time()
ora = seeded_init_(OracleUNet2DConditionModel(**cfg), 1)
with torch.device("meta"):
    nat = NativeUNet2DConditionModel(**cfg)
nat = seeded_init_(nat.to_empty(device=dev), 1)
if "stream-off" in sys.argv[1:]:
    nat.set_residual_stream(False)
if with_lora:
    from hcp_diffusion_amd.trainer import NativeTrainer
    from oracle.lora_ref import wrap_lora
    from oracle.make_golden import sd15_lora_init_
    PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
    rank = 8 if sd15 else 16
    ora.requires_grad_(False)
    wrap_lora(ora, PATS, rank=rank)
    sd15_lora_init_([(n, p) for n, p in ora.named_parameters() if "lora_block_" in n])
    _tr = NativeTrainer(nat, [dict(layers=PATS, rank=rank)], lr=1e-4)
    sd15_lora_init_([(n, p) for n, p in nat.named_parameters() if "lora_block_" in n])
    _tr.bucket.pack()
    assert sorted(n for n, _ in ora.named_parameters() if "lora_block_" in n) == sorted(n for n, _ in nat.named_parameters() if "lora_block_" in n)
if sd15:
    x0, ehs, noise, t = sd15_b4_inputs(); added = None
else:
    x0, ehs, noise, t, added = sdxl_b2_inputs()
if smoke:
    g2 = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, 4, 16, 16, generator=g2); ehs = torch.randn(2, 24, 64, generator=g2); noise = torch.randn(2, 4, 16, 16, generator=g2)
    added = dict(text_embeds=torch.randn(2, 64, generator=g2), time_ids=added["time_ids"])
xt = add_noise(x0, noise, t, ddpm_alphas_cumprod())
kw = {"added_cond_kwargs": added} if added else {}

o_mod, n_mod = dict(ora.named_modules()), dict(nat.named_modules())
names = [n for n, m in o_mod.items() if type(m).__name__ in ("ResnetBlock2D", "Transformer2DModel", "Downsample2D", "Upsample2D", "TimestepEmbedding")]
# the unmodified end-to-end forwards — native, and the oracle under autocast — with the CUMULATIVE error at every block boundary
bnames = (["conv_in"] + [f"down_blocks.{i}" for i in range(len(ora.down_blocks))] + ["mid_block"] +
          [f"up_blocks.{i}" for i in range(len(ora.up_blocks))] + ["conv_norm_out"])
bound = {"fp32": {}, "native": {}, "autocast": {}}


def bhooks(mods, tag, native):
    hs = []
    for n in bnames:
        def f(m, a, out, n=n):
            y = out[0] if isinstance(out, tuple) else out
            y = (y.permute(0, 3, 1, 2) if native else y).float().cpu()
            bound[tag][n] = torch.nn.functional.silu(y) if (n == "conv_norm_out" and not native) else y     # the native module fuses the SiLU
        hs.append(mods[n].register_forward_hook(f))
    return hs


rec = {}
hooks = bhooks(o_mod, "fp32", False) + [o_mod[n].register_forward_hook(lambda m, a, out, n=n: rec.__setitem__(n, (a, out))) for n in names]
with torch.no_grad():
    pred32 = ora(xt, t, ehs, **kw).sample
for h in hooks:
    h.remove()
print(f"oracle fp32 forward + {len(names)} module records: {time.time() - t0:.1f} s", flush=True)

nhwc = lambda y: (y.permute(0, 2, 3, 1).contiguous() if y.dim() == 4 else y).to(torch.bfloat16).to(dev)
got = {}


def pre(mod, args, kwargs, n):
    a = rec[n][0]
    x = a[0]
    if "skip" in kwargs and kwargs["skip"] is not None:                    # up-block resnet: the oracle's input is cat([h, skip])
        c_skip = kwargs["skip"].shape[-1]
        return (nhwc(x[:, :x.shape[1] - c_skip]),) + tuple(args[1:]), dict(kwargs, skip=nhwc(x[:, x.shape[1] - c_skip:]))
    return (nhwc(x),) + tuple(args[1:]), kwargs


hooks = []
for n in names:
    hooks.append(n_mod[n].register_forward_pre_hook(lambda m, a, k, n=n: pre(m, a, k, n), with_kwargs=True))
    def keep(m, a, out, n=n):
        y = out[0] if isinstance(out, tuple) else out
        got[n] = (y.permute(0, 3, 1, 2) if y.dim() == 4 else y).float().cpu()
    hooks.append(n_mod[n].register_forward_hook(keep))
nkw = {"added_cond_kwargs": {k: v.to(dev) for k, v in added.items()}} if added else {}
with torch.no_grad():
    nat(xt.to(dev), t.to(dev), ehs.to(dev), **nkw)
for h in hooks:
    h.remove()
with torch.no_grad():
    hs = bhooks(n_mod, "native", True)
    pred_n = nat(xt.to(dev), t.to(dev), ehs.to(dev), **nkw).sample.float().cpu()
    [h.remove() for h in hs]
    hs = bhooks(o_mod, "autocast", False)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        pred_a = ora(xt, t, ehs, **kw).sample.float()
    [h.remove() for h in hs]
print("\ncumulative error at the block boundaries (own end-to-end forward vs the fp32 oracle):")
for n in bnames:
    r = bound["fp32"][n]
    en, ea = ((bound["native"][n] - r).norm() / r.norm()).item(), ((bound["autocast"][n] - r).norm() / r.norm()).item()
    print(f"  {n:18s} native {en:.3e}  autocast {ea:.3e}  ratio {en / ea:.2f}")
rel = lambda a: ((a - pred32).norm() / pred32.norm()).item()
print(f"END TO END ({'with LoRA: (b) is the reference mixed-precision mode' if with_lora else 'no LoRA'}): native rel-L2 {rel(pred_n):.3e}, autocast-oracle rel-L2 {rel(pred_a):.3e}, ratio {rel(pred_n) / rel(pred_a):.2f}", flush=True)
print(f"native forward with substituted inputs: {time.time() - t0:.1f} s", flush=True)

rows = []
for n in names:
    a, ref = rec[n]
    ref = ref[0] if isinstance(ref, tuple) else ref
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ins = [v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else
               (tuple(u.to(torch.bfloat16) for u in v) if isinstance(v, tuple) else v) for v in a]
        ya = o_mod[n](*ins)
    ya = (ya[0] if isinstance(ya, tuple) else ya).float()
    e_nat = ((got[n] - ref).norm() / ref.norm()).item()
    e_ac = ((ya - ref).norm() / ref.norm()).item()
    rows.append((n, type(o_mod[n]).__name__, e_nat, e_ac))
    print(f"{n:40s} {type(o_mod[n]).__name__:20s} native {e_nat:.3e}  autocast {e_ac:.3e}  ratio {e_nat / e_ac:.2f}", flush=True)

agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for n, kind, en, ea in rows:
    for key in (kind, ".".join(n.split(".")[:2]) + "|" + kind, "ALL"):
        a = agg[key]; a[0] += en * en; a[1] += ea * ea; a[2] += 1
print("\nclass                                     rms native   rms autocast   ratio   n")
for key in sorted(agg):
    a = agg[key]
    print(f"{key:40s} {(a[0] / a[2]) ** 0.5:.3e}    {(a[1] / a[2]) ** 0.5:.3e}    {(a[0] / a[1]) ** 0.5:.2f}   {a[2]}")
print(f"total {time.time() - t0:.1f} s")
