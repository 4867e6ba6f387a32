"""Full-size parity for the BASELINE.json configurations that round 2 only covered at miniature size (missing #1):

  configs[2]  SD1.5 DreamBooth full fine-tune, 512 px — the step of cfgs/train/examples/DreamBooth.yaml: instance batch (bs 2) +
              class batch (bs 1), ONE accumulated gradient over all 686 parameter tensors (train_ac.py:467-483)
  configs[3]  SDXL-base LoRA r=16 at its REAL shape: bs 2, 1024 px = 128x128 latents, the ENTIRE flat LoRA gradient (41.9 M elements)
  configs[4]  frozen SD1.5 + ControlNet branch, bs 4, 512x512 control image, every branch parameter's gradient (controlnet.py:88-183)

and the reference's own ``Trainer.train_one_step`` trajectory (ten steps, train_ac.py:467-504, recorded by
oracle/make_golden.reference_trainer_trajectory over the fp32 oracle modules with the reference's LoRA code and torch AdamW) that
``NativeTrainer`` must reproduce — on the interpreter here and on the MI355X under ``-m gpu``.

The fp32 oracle values were generated in the build container (oracle/make_golden.py dreambooth | sdxl_b2 | controlnet_b4 | trainer)
and travel as fixtures under tests/golden/.  Tolerances (SURVEY.md §8c, bf16 native vs fp32 oracle): prediction rel-L2 <= 2e-2
(SDXL, 70 transformer blocks deep: 3e-2), loss <= 1e-2 relative, flat gradient cosine >= 0.999 for LoRA (SDXL: >= 0.998 flat and >= 0.985
per tensor, the measured bf16-residual-stream class, see the test); for host-parameter
gradients (no figure in §8c) flat cosine over the sampled elements >= 0.995 and every tensor's own cosine >= 0.97."""
import os

import pytest
import torch
import torch.nn.functional as F

from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.unet_sd15 import SDXL_CONFIG, seeded_init_

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


def _native_full(dev, cfg=None):
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel(**(cfg or {}))
    return seeded_init_(nat.to_empty(device=dev), 1)


def _sketch_compare(gold, named_grads):
    """(flat cosine over all sampled elements, [(name, cosine)] below 0.97, worst relative norm error) of live gradients against
    the golden per-tensor sketches (oracle/make_golden.tensor_sketch)."""
    from oracle.make_golden import sketch_values
    num = da = db = 0.0
    bad, worst_norm = [], 0.0
    for name, g in named_grads:
        sk = gold["sketch"][name]
        ref = sk["q"].double() * sk["scale"]
        got = sketch_values(name, g).double()
        assert got.shape == ref.shape, name
        num += float(ref @ got); da += float(ref @ ref); db += float(got @ got)
        if sk["norm"] > 1e-7:
            cos = float(ref @ got) / max(1e-30, float(ref.norm() * got.norm()))
            if cos < 0.97:
                bad.append((name, round(cos, 4)))
            worst_norm = max(worst_norm, abs(float(g.float().norm()) - sk["norm"]) / sk["norm"])
    return num / (da * db) ** 0.5, bad, worst_norm


@pytest.mark.gpu
def test_dreambooth_full_size_two_dataset_step_vs_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import dreambooth_inputs
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sd15_dreambooth_b2_oracle.pt"))
    nat = _native_full(dev)
    tr = NativeTrainer(nat, None, lr=1e-6, train_cfg=[dict(layers=[""], lr=1e-6)])          # DreamBooth.yaml:6-10
    named = list(nat.named_parameters())
    assert [n for n, _ in named] == g["names"] and len(named) == 686
    data = dreambooth_inputs()
    cur = {"i": 0}

    def make_noise(lat):
        d = data[cur["i"]]
        return K.add_noise(lat, d["noise"].to(dev), d["t"].to(dev), tr.acp), d["noise"].to(dev), d["t"].to(dev)
    tr.make_noise = make_noise
    losses = []
    for i, d in enumerate(data):                       # train_ac.py:469-482: every dataset's backward accumulates
        cur["i"] = i
        with torch.no_grad():
            pred = nat(make_noise(d["x0"].to(dev))[0], d["t"].to(dev), d["ehs"].to(dev)).sample.cpu()
        assert ((pred - g["preds"][i]).norm() / g["preds"][i].norm()).item() < 2e-2
        tr.loss_weight = d["loss_weight"]
        losses.append(tr.forward_backward(d["x0"].to(dev), d["ehs"].to(dev)).item())
    for ln, lo in zip(losses, g["losses"]):
        assert abs(ln - lo) / lo < 1e-2, (losses, g["losses"])
    cos, bad, worst_norm = _sketch_compare(g, [(n, p.grad) for n, p in named])
    total = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for _, p in named)))
    print(f"[dreambooth] flat cosine {cos:.5f}, tensors below 0.97: {len(bad)}, worst norm error {worst_norm:.3f}, |g| {total:.5f} vs {g['grad_norm']:.5f}")
    assert cos > 0.995 and not bad, (cos, bad[:10])
    assert abs(total - g["grad_norm"]) / g["grad_norm"] < 2e-2
    # one clip + AdamW step over the 859.5 M-element bucket leaves finite parameters and cleared gradients
    tr.all_reduce(); tr.optimizer_step()
    hb = tr.host_buckets[0].bucket
    assert torch.isfinite(hb.params).all().item() and hb.grads.abs().max().item() == 0.0


# native error / reference-under-autocast error (both against the fp32 oracle), as measured on MI355X + margin; see the test body
# measured r4: 1.45 / 2.18 (median class 1.59) / 1.29 / 2.61; r5 (split T/U run): 1.46 / 2.15 / 1.29 / 2.96;
# r6 with the (hi | lo) residual stream (on by default for SDXL's stacks): 1.38-1.41 / 1.92-1.99 (median 1.49-1.52) / 1.20-1.21 / 1.79-1.93.
# These are the ratios of THIS fixture's input draw: tools/diag/sdxl_grad_draws.py repeats the comparison on six other draws — prediction
# 0.90 ... 1.05, flat gradient 0.65 ... 1.17, means of the seven 1.01 / 1.02 (profiles/r6_diag_sdxl_grad_draws.txt).
R_SDXL_FLAT, R_SDXL_CLASS, R_SDXL_PRED, R_SDXL_TENSOR = 1.6, 2.4, 1.35, 2.4


@pytest.mark.gpu
def test_sdxl_full_size_b2_1024px_full_lora_gradient_vs_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import dequantize_grads, sd15_lora_init_, sdxl_b2_inputs
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sdxl_full_b2_oracle.pt"))
    nat = _native_full(dev, SDXL_CONFIG)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=16)], lr=1e-4)
    by_name = {n: p for n, p in nat.named_parameters() if "lora_block_" in n}
    assert sorted(by_name) == sorted(g["grad_names"]) and tr.bucket.numel == 41_861_120
    lora_named = [(n, by_name[n]) for n in g["grad_names"]]
    sd15_lora_init_(lora_named)
    tr.bucket.pack()
    x0, ehs, noise, t, added = sdxl_b2_inputs()
    added = {k: v.to(dev) for k, v in added.items()}
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=added).sample.cpu()
    ref_pred = g["pred"].float()
    assert ((pred - ref_pred).norm() / ref_pred.norm()).item() < 3e-2           # 70 transformer blocks deep in bf16
    loss = tr.forward_backward(x0.to(dev), ehs.to(dev), None, added).item()
    assert abs(loss - g["loss"]) / g["loss"] < 1e-2
    flat = torch.cat([p.grad.detach().float().flatten().cpu() for _, p in lora_named])
    ref = dequantize_grads(g["grad_q"], g["grad_scales"], lora_named)
    cos = (flat.double() @ ref.double() / (flat.double().norm() * ref.double().norm())).item()
    worst, off = (1.0, ""), 0
    for n, p in lora_named:                            # every one of the 1400 tensors on its own
        a, b = flat[off:off + p.numel()].double(), ref[off:off + p.numel()].double(); off += p.numel()
        c = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))
        worst = min(worst, (c, n))
    print(f"[sdxl b2 1024px] LoRA gradient cosine {cos:.5f} over {flat.numel()} elements, worst tensor {worst[0]:.4f} ({worst[1]}), "
          f"norm {flat.norm().item():.5f} vs {g['grad_norm']:.5f}")
    # Calibrated gate: tests/golden/sdxl_b2_autocast_calibration.pt holds the distance of the REFERENCE's own
    # execution mode — the same oracle graph under torch.autocast(bfloat16), train_ac.py:449 — from this fp32 fixture (oracle/make_golden.py
    # sdxl_b2_autocast: flat cosine 0.99899, worst tensor 0.99667, prediction rel-L2 2.03e-2).  The native step must stay within a stated
    # multiple of THAT error, per (resolution block, layer kind, factor) class, flat, and on the prediction, instead of hand-set
    # absolute figures.  Round 5 (profiles/r5_ab_t_split.md, profiles/r5_diag_*): the excess over that baseline is NOT the bf16
    # rank-r LoRA intermediates (kernels.T_SPLIT carries them to 16 mantissa bits: ratios unchanged) and not the kernels' arithmetic
    # (module by module the native error is 0.94-1.00 of plain bf16 autocast's; end to end without LoRA 1.01): the reference's LoRA layers
    # compute mm(x, W^T) + fp32 bias (lora_layers_patch.py:50-55), which PROMOTES the outputs of to_out.0 / ff.net.0.proj / ff.net.2 — and
    # with them the transformer blocks' residual stream and its gradient — to fp32 under autocast, while plain autocast and the native path
    # round the stream to bf16 at each of the 210 residual adds.  The gates below are multiples of that stricter, mixed-precision baseline.
    from oracle.make_golden import lora_tensor_class
    cal = torch.load(os.path.join(GOLD, "sdxl_b2_autocast_calibration.pt"))
    assert cal["names"] == g["grad_names"]
    acc, off = {}, 0
    for n, p in lora_named:
        a, b = flat[off:off + p.numel()].double(), ref[off:off + p.numel()].double(); off += p.numel()
        c_ = acc.setdefault(lora_tensor_class(n), [0.0, 0.0, 0.0]); c_[0] += float(a @ b); c_[1] += float(a @ a); c_[2] += float(b @ b)
    ratios = {k: (1.0 - v[0] / (v[1] * v[2]) ** 0.5) / max(1.0 - cal["class_cos"][k], 1e-9) for k, v in acc.items()}
    worst_cls = max(ratios.items(), key=lambda kv: kv[1])
    r_flat = (1.0 - cos) / (1.0 - cal["flat_cos"])
    r_pred = ((pred - ref_pred).norm() / ref_pred.norm()).item() / cal["pred_rel"]
    r_worst_tensor = (1.0 - worst[0]) / (1.0 - cal["per_tensor_cos"].min().item())
    print(f"[sdxl b2 calibration] (1 - cos) native / autocast: flat {r_flat:.2f}, worst class {worst_cls[1]:.2f} ({worst_cls[0]}), median class "
          f"{sorted(ratios.values())[len(ratios) // 2]:.2f}, worst tensor {r_worst_tensor:.2f}; prediction rel-L2 ratio {r_pred:.2f}")
    assert r_flat < R_SDXL_FLAT and worst_cls[1] < R_SDXL_CLASS and r_pred < R_SDXL_PRED and r_worst_tensor < R_SDXL_TENSOR
    assert abs(flat.norm().item() - g["grad_norm"]) / g["grad_norm"] < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("draw", [1, 2])
def test_sdxl_full_size_b2_other_input_draws_vs_golden(draw):
    """configs[3] on OTHER input draws than the fixture above (round 6): tests/golden/sdxl_b2_draw<d>_oracle.pt holds, for
    latents / prompt states / noise / timesteps from seed 1000 + d, the fp32 oracle's prediction, a seeded 2 M-element sketch of its flat LoRA
    gradient and the distances of the REFERENCE's mixed-precision LoRA mode (the same oracle under autocast) from both — written by
    tools/diag/sdxl_grad_draws.py save=... on the GPU box's host cores.  The first fixture's ratios (prediction 1.20, gradient 1.38) are
    the worst of seven draws; on these two the native step must be within 1.2 x the reference mode's error on BOTH counts (measured:
    0.96 / 0.65 and 0.97 / 0.82) — the bar VERDICT r5 set for the flat ratio."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import sd15_lora_init_, sdxl_b2_draw_inputs
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, f"sdxl_b2_draw{draw}_oracle.pt"))
    nat = _native_full(dev, SDXL_CONFIG)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=16)], lr=1e-4)
    named = sorted((n, p) for n, p in nat.named_parameters() if "lora_block_" in n)
    assert [n for n, _ in named] == g["names"]
    sd15_lora_init_(named)
    tr.bucket.pack()
    x0, ehs, noise, t, added = sdxl_b2_draw_inputs(draw)
    added = {k: v.to(dev) for k, v in added.items()}
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=added).sample.float().cpu()
    ref_pred = g["pred"].float()
    r_pred = ((pred - ref_pred).norm() / ref_pred.norm()).item() / g["pred_rel_ref"]
    tr.forward_backward(x0.to(dev), ehs.to(dev), None, added)
    flat = torch.cat([p.grad.detach().flatten().double().cpu() for _, p in named])
    idx = torch.randint(0, flat.numel(), (g["sketch_n"],), generator=torch.Generator().manual_seed(g["sketch_seed"]))
    sk, ref = flat[idx], g["sketch_fp32"].double() * g["sketch_scale"]
    cos = float(sk @ ref / (sk.norm() * ref.norm()))
    r_grad = (1.0 - cos) / (1.0 - g["ref_sketch_cos"])
    print(f"[sdxl b2 draw {draw}] native / reference-mode error: prediction {r_pred:.2f}, LoRA gradient (1 - cos, 2 M-element sketch) {r_grad:.2f} "
          f"(reference mode: pred rel-L2 {g['pred_rel_ref']:.3e}, sketch cos {g['ref_sketch_cos']:.5f}); |g| {flat.norm().item():.5f} vs {g['grad_norm']:.5f}")
    assert r_pred < 1.2 and r_grad < 1.2
    assert abs(flat.norm().item() - g["grad_norm"]) / g["grad_norm"] < 2e-2


@pytest.mark.gpu
def test_controlnet_full_size_b4_branch_gradients_vs_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hcp_diffusion_amd.controlnet import make_controlnet
    from oracle.make_golden import controlnet_b4_inputs, controlnet_init_
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sd15_controlnet_b4_oracle.pt"))
    nat = _native_full(dev)
    torch.manual_seed(3)
    plug = make_controlnet(nat)                        # deep copy of the host encoder (same seeded weights as the oracle's copy)
    controlnet_init_(plug)
    named = list(plug.named_parameters())
    assert sorted(n for n, _ in named) == sorted(g["names"]) and sum(p.numel() for _, p in named) > 360_000_000
    # (every cond_head / zero-conv parameter is re-drawn BY NAME on the CPU by controlnet_init_, exactly as on the oracle side; the
    #  encoder copy carries the host's seeded weights: nothing depends on a device generator)
    tr = NativeTrainer(nat, None, lr=1e-5, plugins=[(plug, 1e-5)])
    x0, ehs, noise, t, cond = controlnet_b4_inputs()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    loss = tr.forward_backward(x0.to(dev), ehs.to(dev), None, None, dict(cond=cond.to(dev))).item()
    assert abs(loss - g["loss"]) / g["loss"] < 1e-2, (loss, g["loss"])
    cos, bad, worst_norm = _sketch_compare(g, [(n, p.grad) for n, p in named])
    total = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for _, p in named)))
    print(f"[controlnet b4] flat cosine {cos:.5f}, tensors below 0.97: {len(bad)}, worst norm error {worst_norm:.3f}, |g| {total:.5f} vs {g['grad_norm']:.5f}")
    assert cos > 0.995 and not bad, (cos, bad[:10])
    assert abs(total - g["grad_norm"]) / g["grad_norm"] < 2e-2
    with torch.no_grad():                              # prediction with the branch attached
        for feeder in nat.input_feeder:
            feeder(dict(cond=cond.to(dev)))
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev)).sample.cpu()
    assert ((pred - g["pred"]).norm() / g["pred"].norm()).item() < 2e-2


def test_native_trainer_reproduces_the_reference_trainer_trajectory(backend):
    """tests/golden/ref_trainer_trajectory.pt: losses and final LoRA factors of ten steps of the reference's own train_one_step (fp32,
    oracle modules, reference LoRA, torch AdamW).  NativeTrainer on the same data, noise and timesteps: bf16 pipeline vs fp32 —
    every loss within 1e-2 relative; the LoRA UPDATE (final - initial: ten Adam steps of lr 1e-3) cosine >= 0.98 flat."""
    from hcp_diffusion_amd.text_encoder import NativeCLIPTextModel
    from oracle.clip_ref import OracleCLIPTextModel
    g = torch.load(os.path.join(GOLD, "ref_trainer_trajectory.pt"))
    dev = backend.device
    u = seeded_init_(NativeUNet2DConditionModel(**g["unet_cfg"]), 1).to(dev)
    te = NativeCLIPTextModel(**g["te_cfg"]); te.load_state_dict(seeded_init_(OracleCLIPTextModel(**g["te_cfg"]), 2).state_dict()); te.to(dev)
    tr = NativeTrainer(u, [dict(layers=PATS, rank=4, lr=g["lr"])], lr=g["lr"], weight_decay=g["weight_decay"], text_encoder=te)
    assert set(tr.lora_group.plugin_dict) == set(g["lora_init"])
    with torch.no_grad():
        for path, blk in tr.lora_group.plugin_dict.items():
            blk.layer.W_down.copy_(g["lora_init"][path][0]); blk.layer.W_up.copy_(g["lora_init"][path][1])
    tr.bucket.pack()
    step = {"i": 0}

    def make_noise(lat):                               # the draws of the reference's torch CPU generator, replayed
        noise, t = g["drawn"][step["i"]]
        return K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev)
    tr.make_noise = make_noise
    losses = []
    for i, d in enumerate(g["data"]):
        step["i"] = i
        losses.append(tr.train_one_step(d["img"].to(dev), None, prompt_ids=d["prompt"].to(dev)).item())
    for a, b in zip(losses, g["losses"]):
        assert abs(a - b) / b < 1e-2, (losses, g["losses"])
    assert losses[0] != losses[-1]
    upd_n, upd_r = [], []
    for path, blk in tr.lora_group.plugin_dict.items():
        for k, p in enumerate((blk.layer.W_down, blk.layer.W_up)):
            upd_n.append((p.detach().cpu() - g["lora_init"][path][k]).flatten())
            upd_r.append((g["lora_final"][path][k] - g["lora_init"][path][k]).flatten())
    upd_n, upd_r = torch.cat(upd_n), torch.cat(upd_r)
    cos = F.cosine_similarity(upd_n, upd_r, dim=0).item()
    print(f"[ref trajectory] losses native {losses[0]:.5f}..{losses[-1]:.5f} reference {g['losses'][0]:.5f}..{g['losses'][-1]:.5f}; update cosine {cos:.4f}")
    assert cos > 0.98 and abs(upd_n.norm().item() - upd_r.norm().item()) / upd_r.norm().item() < 5e-2
