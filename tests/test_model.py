"""Model-level parity: the native UNet / LoRA training step (HIP kernels) vs the CPU oracle on the same seeded inputs.

Tolerances: bf16 native vs fp32 oracle — relative L2 error of `.sample` <= 2e-2 end to end, LoRA
gradient cosine >= 0.995 (flat, all layers), loss within 2e-2 relative.
`backend` = "emu": kernels interpreted on the CPU (TINY config);  "gpu" (-m gpu): gfx950 library, TINY and full SD1.5.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.lora import LoraHipLayer, make_lora
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.lora_ref import wrap_lora
from oracle.unet_sd15 import (MICRO_CONFIG, OracleUNet2DConditionModel, SD15_CONFIG, SDXL_CONFIG, TINY_CONFIG, TINY_SDXL_CONFIG,
                              add_noise, ddpm_alphas_cumprod, seeded_init_)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


def test_native_unet_names_match_reference_struct():
    ref = json.load(open(os.path.join(GOLD, "sd15_struct.json")))["shapes"]
    with torch.device("meta"):
        m = NativeUNet2DConditionModel()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref
    leaves = dict(m.named_modules())
    assert isinstance(leaves["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"], torch.nn.Linear)
    assert isinstance(leaves["down_blocks.0.resnets.0.conv1"], torch.nn.Conv2d)
    assert type(leaves["down_blocks.0"]).__name__ == "CrossAttnDownBlock2D" and type(leaves["up_blocks.0"]).__name__ == "UpBlock2D"


def _pair(cfg, dev, rank=4):
    torch.manual_seed(0)
    ora = seeded_init_(OracleUNet2DConditionModel(**cfg), 1)
    nat = NativeUNet2DConditionModel(**cfg)
    nat.load_state_dict(ora.state_dict())
    nat.to(dev)
    return ora, nat


def test_tiny_forward_vs_oracle_and_golden(backend):
    g = torch.load(os.path.join(GOLD, "tiny_unet_oracle.pt"))
    ora, nat = _pair(TINY_CONFIG, backend.device)
    xt = add_noise(g["x0"], g["noise"], g["t"], ddpm_alphas_cumprod())
    with torch.no_grad():
        yo = ora(xt, g["t"], g["ehs"]).sample
        yn = nat(backend.to(xt), backend.to(g["t"]), backend.to(g["ehs"])).sample.cpu()
    assert yn.dtype == torch.float32 and yn.shape == yo.shape
    assert ((yn - yo).norm() / yo.norm()).item() < 2e-2


@pytest.mark.parametrize("B,h,w,L", [(1, 8, 12, 77), (3, 12, 8, 9), (2, 16, 8, 154), (1, 96, 64, 77)])
def test_forward_ragged_shapes_vs_oracle(backend, B, h, w, L):
    """Aspect-ratio buckets give non-square latents (data/bucket.py), batch sizes of 1 and odd counts, prompts of 1 / N x 77 tokens
    (tokenizer_repeats): the native forward against the oracle on shapes that are not multiples of any tile size."""
    if not backend.is_gpu and h * w > 200:
        pytest.skip("large shape: GPU only")
    ora, nat = _pair(MICRO_CONFIG, backend.device)
    g = torch.Generator().manual_seed(h * w + L)
    x = torch.randn(B, 4, h, w, generator=g); t = torch.randint(0, 1000, (B,), generator=g); ehs = torch.randn(B, L, 32, generator=g)
    with torch.no_grad():
        yo = ora(x, t, ehs).sample
        yn = nat(backend.to(x), backend.to(t), backend.to(ehs)).sample.cpu()
    assert yn.shape == yo.shape == (B, 4, h, w)
    assert ((yn - yo).norm() / yo.norm()).item() < 2e-2


def test_tiny_forward_with_encoder_attention_mask(backend):
    """The call contract's `encoder_attention_mask=[B,L]` (models/wrapper.py:22-29; the reference pads L to a multiple of 8)."""
    ora, nat = _pair(TINY_CONFIG, backend.device)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([5, 900]); ehs = torch.randn(2, 80, 64, generator=g)
    mask = torch.ones(2, 80); mask[0, 30:] = 0; mask[1, 77:] = 0
    with torch.no_grad():
        yo = ora(x, t, ehs, encoder_attention_mask=mask).sample
        yo_nomask = ora(x, t, ehs).sample
        yn = nat(backend.to(x), backend.to(t), backend.to(ehs), encoder_attention_mask=backend.to(mask)).sample.cpu()
    assert ((yn - yo).norm() / yo.norm()).item() < 2e-2
    assert ((yo - yo_nomask).norm() / yo.norm()).item() > 5e-2          # the mask matters for this input


def _train_step_pair(cfg, backend, rank, shape, ctx_len, ctx_dim, seed=42, pooled_dim=None, loss_cfg=None, stream=None, geglu_epilogue=False):
    dev = backend.device
    ora, nat = _pair(cfg, dev)
    if stream is not None:
        nat.set_residual_stream(stream)
    if geglu_epilogue:
        nat.set_geglu_epilogue(True)
    ora.requires_grad_(False)
    wr = wrap_lora(ora, PATS, rank=rank)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=rank)], lr=1e-3, loss_cfg=loss_cfg)
    assert sorted(k for k in ora.state_dict() if "lora" in k) == sorted(k for k in nat.state_dict() if "lora" in k)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for path, w in wr.items():
            blk = tr.lora_group.plugin_dict[path]
            w.lora_block_0.layer.W_up.copy_(torch.randn(w.lora_block_0.layer.W_up.shape, generator=gen) * 0.05)
            blk.layer.W_down.copy_(w.lora_block_0.layer.W_down); blk.layer.W_up.copy_(w.lora_block_0.layer.W_up)
    tr.bucket.pack()
    g2 = torch.Generator().manual_seed(seed)
    x0 = torch.randn(*shape, generator=g2); ehs = torch.randn(shape[0], ctx_len, ctx_dim, generator=g2)
    noise = torch.randn(*shape, generator=g2); t = torch.randint(0, 1000, (shape[0],), generator=g2).long()
    added = None
    if pooled_dim:                                   # SDXL call contract (reference models/wrapper.py:66)
        added = dict(text_embeds=torch.randn(shape[0], pooled_dim, generator=g2),
                     time_ids=torch.tensor([[shape[2] * 8.0, shape[3] * 8.0, 0.0, 16.0, shape[2] * 8.0, shape[3] * 8.0]] * shape[0]))
    pred = ora(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs, added_cond_kwargs=added).sample
    if loss_cfg:
        from oracle.loss_ref import get_loss
        loss_o = get_loss(pred, noise, None, kind=loss_cfg["type"], timesteps=t, alphas_cumprod=ddpm_alphas_cumprod(), gamma=loss_cfg["gamma"])
    else:
        loss_o = F.mse_loss(pred, noise)
    loss_o.backward()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    loss_n = tr.forward_backward(x0.to(dev), ehs.to(dev), None, {k: v.to(dev) for k, v in added.items()} if added else None)
    go = torch.cat([p.grad.flatten() for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)])
    return loss_o.item(), loss_n.item(), go, tr, wr


def test_tiny_lora_train_step_vs_oracle(backend):
    lo, ln, go, tr, wr = _train_step_pair(TINY_CONFIG, backend, 4, (2, 4, 8, 8), 77, 64)
    assert abs(lo - ln) / abs(lo) < 2e-2
    assert F.cosine_similarity(go, tr.bucket.grads.cpu(), dim=0).item() > 0.999
    # fused clip + AdamW + re-pack against torch's clip_grad_norm_ + AdamW on the oracle's (near identical) gradients
    params = [p for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)]
    for p, gslice in zip(params, torch.split(tr.bucket.grads.cpu(), [p.numel() for p in params])):
        p.grad = gslice.view_as(p).clone()          # same gradients both sides: isolates the optimizer arithmetic
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-3)
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    tr.all_reduce(); tr.optimizer_step()
    po = torch.cat([p.detach().flatten() for p in params])
    assert ((po - tr.bucket.params.cpu()).abs().max() / po.abs().max()).item() < 1e-5
    assert tr.bucket.grads.abs().max().item() == 0.0


def test_tiny_lora_train_step_min_snr_loss(backend):
    """train.loss.criterion = MinSNRLoss(gamma=5) (hcpdiff/loss/min_snr_loss.py): loss and LoRA gradients vs the oracle UNet
    under oracle/loss_ref.get_loss (itself pinned to the reference classes); the weights differ from 1 for this batch."""
    lo, ln, go, tr, _ = _train_step_pair(MICRO_CONFIG, backend, 4, (2, 4, 8, 8), 77, 32, loss_cfg=dict(type="min_snr", gamma=5.0))
    assert abs(lo - ln) / abs(lo) < 2e-2
    assert F.cosine_similarity(go, tr.bucket.grads.cpu(), dim=0).item() > 0.999
    with pytest.raises(ValueError):
        NativeTrainer(tr.unet, None, train_cfg=[dict(layers=[""])], loss_cfg=dict(type="huber"))


def test_sdxl_structure_matches_public_config():
    """SDXL-base: 2.567 B parameters in 1680 tensors; attentions at down_blocks.1-2 / mid / up_blocks.0-1 (the layout the
    reference's converters assume, hcpdiff/tools/lora_convert.py:116-186); linear proj_in/out; same names as the oracle."""
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel(**SDXL_CONFIG)
        ora = OracleUNet2DConditionModel(**SDXL_CONFIG)
    sn = {k: tuple(v.shape) for k, v in nat.state_dict().items()}
    assert sn == {k: tuple(v.shape) for k, v in ora.state_dict().items()}
    assert len(sn) == 1680 and sum(torch.Size(v).numel() for v in sn.values()) == 2_567_463_684
    assert sn["add_embedding.linear_1.weight"] == (1280, 2816) and sn["down_blocks.1.attentions.0.proj_in.weight"] == (640, 640)
    assert not hasattr(nat.down_blocks[0], "attentions") and not hasattr(nat.up_blocks[2], "attentions")
    assert [len(a.transformer_blocks) for a in nat.up_blocks[0].attentions] == [10, 10, 10]
    assert len(nat.mid_block.attentions[0].transformer_blocks) == 10 and len(nat.down_blocks[1].attentions[0].transformer_blocks) == 2
    assert nat.down_blocks[2].attentions[0].transformer_blocks[0].attn1.heads == 20
    with pytest.raises(ValueError):
        NativeUNet2DConditionModel(**TINY_SDXL_CONFIG)(torch.zeros(1, 4, 8, 8), torch.zeros(1).long(), torch.zeros(1, 77, 64))


@pytest.mark.parametrize("t_split", [False, True])
def test_tiny_sdxl_lora_train_step_vs_oracle(backend, t_split, monkeypatch):
    """SDXL-structured miniature (3 levels, DownBlock2D first, transformer depth 1/1/2, head_dim 64, linear projections,
    text_time additional embedding, rank-16 LoRA as in BASELINE.json configs[3]).  t_split: the opt-in (hi | lo) form of the rank-r
    intermediates (kernels.T_SPLIT) through every path that carries them — fused linear / group GEMMs, GEGLU backward epilogue, the
    batched cross-attention K|V projection."""
    monkeypatch.setattr(K, "T_SPLIT", t_split)
    lo, ln, go, tr, wr = _train_step_pair(TINY_SDXL_CONFIG, backend, 16, (2, 4, 8, 8), 77, 64, pooled_dim=64)
    cb = getattr(tr.unet, "_ctx_batch", None)            # (key, CtxBatch or None, groups)
    assert cb is not None and cb[1] is not None and cb[1].split == t_split
    assert abs(lo - ln) / abs(lo) < 2e-2
    assert F.cosine_similarity(go, tr.bucket.grads.cpu(), dim=0).item() > 0.999
    # three rank-16 blocks do not fit one 32-wide slot group: self-attention runs as q alone + k|v together (the pre-scaled-Q kernels)
    a1 = next(m for n, m in tr.unet.named_modules() if n.endswith("attn1"))
    built = sorted(len(k) for k, (g, _) in a1._groups.items() if g is not None)
    assert built == [1, 2] and any(g is None and len(k) == 3 for k, (g, _) in a1._groups.items())


def test_geglu_epilogue_option_train_step(backend, monkeypatch):
    """unet.set_geglu_epilogue(True): the GEGLU product comes out of the FF projection's GEMM epilogue (ops.linear_geglu) — same step
    within the usual tolerances, with the (hi | lo) stream on top, and the stand-alone geglu_fwd pass is really gone."""
    calls = {"n": 0}
    gf = K.geglu_fwd
    monkeypatch.setattr(K, "geglu_fwd", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), gf(*a, **k))[1])
    lo, ln, go, tr, _ = _train_step_pair(TINY_SDXL_CONFIG, backend, 16, (2, 4, 8, 8), 77, 64, pooled_dim=64, stream=True, geglu_epilogue=True)
    assert calls["n"] == 0
    assert abs(lo - ln) / abs(lo) < 2e-2 and F.cosine_similarity(go, tr.bucket.grads.cpu(), dim=0).item() > 0.999
    lo, ln, go, tr, _ = _train_step_pair(TINY_CONFIG, backend, 4, (2, 4, 8, 8), 77, 64)
    assert calls["n"] > 0                              # the default keeps the two-pass form (measured faster)


def test_native_step_matches_the_reference_mixed_precision_mode_over_input_draws(backend):
    """What tests/test_full_configs.py measures on ONE full-size fixture, here on the SDXL miniature over three input draws: the native step's
    distance from the fp32 oracle (prediction rel-L2, LoRA-gradient 1 - cos) against the distance of the REFERENCE's execution mode — the
    same oracle with the reference-form LoRA layers under torch.autocast(bfloat16), whose mm + fp32 bias keeps the transformer stream in
    fp32.  Round 6 found the full-size fixture's ratios (1.21 / 1.38) to be one draw of quantities that scatter around 1 (DESIGN section 4:
    the prediction is a 2880 -> 4 projection of an error whose coherent part is a handful of numbers); the mean over draws is the
    statement, the single draws are printed."""
    dev = backend.device
    ora, nat = _pair(TINY_SDXL_CONFIG, dev)
    ora.requires_grad_(False)
    wr = wrap_lora(ora, PATS, rank=16)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=16)], lr=1e-3)
    o_named = sorted((n, p) for n, p in ora.named_parameters() if "lora_block_" in n)
    n_named = sorted((n, p) for n, p in nat.named_parameters() if "lora_block_" in n)
    assert [a for a, _ in o_named] == [a for a, _ in n_named]
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for (_, po), (_, pn) in zip(o_named, n_named):
            po.copy_(torch.randn(po.shape, generator=gen) * (0.05 if po.shape[1] == 16 else po.shape[1] ** -0.5))
            pn.copy_(po)
    tr.bucket.pack()
    acp = ddpm_alphas_cumprod()
    rp, rg = [], []
    for draw in range(3):
        g2 = torch.Generator().manual_seed(100 + draw)
        x0 = torch.randn(2, 4, 16, 16, generator=g2); ehs = torch.randn(2, 24, 64, generator=g2); noise = torch.randn(2, 4, 16, 16, generator=g2)
        t = torch.randint(0, 1000, (2,), generator=g2)
        added = dict(text_embeds=torch.randn(2, 64, generator=g2), time_ids=torch.tensor([[128.0, 128.0, 0.0, 0.0, 128.0, 128.0]] * 2))
        xt = add_noise(x0, noise, t, acp)
        res = {}
        for mode in ("fp32", "ref"):
            for _, p in o_named:
                p.grad = None
            if mode == "ref":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    pred = ora(xt, t, ehs, added_cond_kwargs=added).sample
            else:
                pred = ora(xt, t, ehs, added_cond_kwargs=added).sample
            F.mse_loss(pred.float(), noise).backward()
            res[mode] = (pred.detach().float(), torch.cat([p.grad.flatten().double() for _, p in o_named]))
        tr.bucket.grads.zero_()
        tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
        dadd = {k: v.to(dev) for k, v in added.items()}
        with torch.no_grad():
            pn = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=dadd).sample.float().cpu()
        tr.forward_backward(x0.to(dev), ehs.to(dev), None, dadd)
        gn = torch.cat([p.grad.detach().flatten().double().cpu() for _, p in n_named])
        p32, g32 = res["fp32"]
        e = lambda a: ((a - p32).norm() / p32.norm()).item()
        c = lambda a: 1.0 - float(a @ g32 / (a.norm() * g32.norm()))
        rp.append(e(pn) / e(res["ref"][0])); rg.append(c(gn) / c(res["ref"][1]))
    print(f"native / reference-mode error over draws: prediction {[round(v, 2) for v in rp]} mean {sum(rp) / 3:.2f}; gradient (1 - cos) {[round(v, 2) for v in rg]} mean {sum(rg) / 3:.2f}")
    assert sum(rp) / 3 < 1.15 and sum(rg) / 3 < 1.25 and max(rp) < 1.5 and max(rg) < 1.7


def test_hi_lo_residual_stream_train_step(backend, monkeypatch):
    """The transformer blocks' residual stream as a (hi | lo) bf16 pair (Transformer2DModel.hi_lo_stream; "auto" = stacks of >= 2 blocks,
    forced on / off here for every stack): both forms train the SDXL miniature within the usual tolerances, the pair form really runs
    (every to_out / ff.net.2 add and every norm of a block takes the stream entry points, forward and backward), and it is not further
    from the fp32 oracle than the bf16 stream."""
    from hcp_diffusion_amd import ops
    calls = {"lin": 0, "ln": 0, "ln_bwd_lo": 0}
    lin, lnf, lnb = ops.linear_stream, ops.layernorm_fork_stream, K.layernorm_bwd
    monkeypatch.setattr(ops, "linear_stream", lambda *a, **k: (calls.__setitem__("lin", calls["lin"] + 1), lin(*a, **k))[1])
    monkeypatch.setattr(ops, "layernorm_fork_stream", lambda *a, **k: (calls.__setitem__("ln", calls["ln"] + 1), lnf(*a, **k))[1])
    monkeypatch.setattr(K, "layernorm_bwd", lambda *a, **k: (calls.__setitem__("ln_bwd_lo", calls["ln_bwd_lo"] + (1 if k.get("want_lo") else 0)), lnb(*a, **k))[1])
    res = {}
    for mode in (False, True):
        lo, ln, go, tr, _ = _train_step_pair(TINY_SDXL_CONFIG, backend, 16, (2, 4, 8, 8), 77, 64, pooled_dim=64, stream=mode)
        cos = F.cosine_similarity(go.double(), tr.bucket.grads.cpu().double(), dim=0).item()
        assert abs(lo - ln) / abs(lo) < 2e-2 and cos > 0.999
        res[mode] = (abs(lo - ln) / abs(lo), 1.0 - cos)
        if not mode:
            assert calls == {"lin": 0, "ln": 0, "ln_bwd_lo": 0}
    nblk = sum(len(m.transformer_blocks) for m in tr.unet.modules() if hasattr(m, "transformer_blocks"))
    # per block: attn1.to_out + attn2.to_out through linear_stream (ff.net.2 rides in geglu_linear), three norms; all but each stack's first
    # norm return a gradient pair
    nstacks = sum(1 for m in tr.unet.modules() if hasattr(m, "transformer_blocks"))
    assert calls["lin"] == 2 * nblk and calls["ln"] == 3 * nblk and calls["ln_bwd_lo"] == 3 * nblk - nstacks
    print(f"hi|lo stream: loss rel {res[True][0]:.2e} (bf16 stream {res[False][0]:.2e}), 1-cos {res[True][1]:.2e} (bf16 stream {res[False][1]:.2e})")
    assert res[True][1] <= res[False][1] * 1.10


def _full_ft_pair(cfg, backend, shape, ctx_len, ctx_dim, pooled_dim=None, seed=11):
    dev = backend.device
    ora, nat = _pair(cfg, dev)
    tr = NativeTrainer(nat, None, lr=1e-3, train_cfg=[dict(layers=[""], lr=1e-3)])          # DreamBooth.yaml:6-10
    g2 = torch.Generator().manual_seed(seed)
    x0 = torch.randn(*shape, generator=g2); ehs = torch.randn(shape[0], ctx_len, ctx_dim, generator=g2)
    noise = torch.randn(*shape, generator=g2); t = torch.randint(0, 1000, (shape[0],), generator=g2).long()
    added = None
    if pooled_dim:
        added = dict(text_embeds=torch.randn(shape[0], pooled_dim, generator=g2),
                     time_ids=torch.tensor([[shape[2] * 8.0, shape[3] * 8.0, 0.0, 16.0, shape[2] * 8.0, shape[3] * 8.0]] * shape[0]))
    pred = ora(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs, added_cond_kwargs=added).sample
    loss_o = F.mse_loss(pred, noise)
    loss_o.backward()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    batch = (x0.to(dev), ehs.to(dev), None, {k: v.to(dev) for k, v in added.items()} if added else None)
    loss_n = tr.forward_backward(*batch)
    return ora, nat, tr, loss_o.item(), loss_n.item(), batch


@pytest.mark.parametrize("cfg_name", ["sd15", "sdxl"])
def test_tiny_full_finetune_step_vs_oracle(backend, cfg_name):
    """Every UNet parameter trainable (reference cfgs/train/examples/DreamBooth.yaml:6-10): weight / bias / norm-affine
    gradients of all layers vs fp32 autograd of the oracle, then clip + AdamW + bf16 operand refresh."""
    cfg, extra = (MICRO_CONFIG, {}) if cfg_name == "sd15" else (TINY_SDXL_CONFIG, dict(pooled_dim=64))
    ora, nat, tr, lo, ln, batch = _full_ft_pair(cfg, backend, (2, 4, 8, 8), 24, cfg["cross_attention_dim"], **extra)
    assert abs(lo - ln) / abs(lo) < 2e-2
    po = dict(ora.named_parameters())
    hb = tr.host_buckets[0].bucket
    assert len(hb.named) == len(po) and hb.numel >= sum(p.numel() for p in po.values())
    num = den_a = den_b = 0.0
    bad = []
    for name, p in nat.named_parameters():
        go, gn = po[name].grad, p.grad.cpu()
        assert gn.shape == go.shape
        num += (go * gn).sum().item(); den_a += go.norm().item() ** 2; den_b += gn.norm().item() ** 2
        cos = F.cosine_similarity(go.flatten(), gn.flatten(), dim=0).item()
        if go.norm().item() > 1e-6 and cos < 0.97:
            bad.append((name, round(cos, 4)))
    assert num / (den_a * den_b) ** 0.5 > 0.995          # all gradients, flat
    assert not bad, bad                                   # and each tensor on its own (bf16 pipeline vs fp32 oracle)
    # optimizer: torch clip + AdamW on the oracle with the NATIVE gradients, vs the fused kernel; then the refreshed
    # bf16 operands must reproduce the updated model's forward
    params = list(po.values())
    for name, p in nat.named_parameters():
        po[name].grad = p.grad.cpu().contiguous().clone()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-3).step()
    tr.all_reduce(); tr.optimizer_step()
    for name, p in nat.named_parameters():
        assert ((po[name].detach() - p.detach().cpu()).abs().max() / po[name].abs().max().clamp_min(1e-6)).item() < 1e-5, name
    assert hb.grads.abs().max().item() == 0.0
    x0, ehs, _, added = batch
    noise_t = tr.make_noise(x0)
    with torch.no_grad():
        kw = dict(added_cond_kwargs=added) if added else {}
        yn = nat(noise_t[0], noise_t[2], ehs, **kw).sample.cpu()
        kwo = dict(added_cond_kwargs={k: v.cpu() for k, v in added.items()}) if added else {}
        yo = ora(noise_t[0].cpu(), noise_t[2].cpu(), ehs.cpu(), **kwo).sample
    assert ((yn - yo).norm() / yo.norm()).item() < 2e-2


def test_lora_layer_api_surface(backend):
    """The reference-facing surface of seam 2: wrap_model returns {path: block}, container replaces the host in its
    parent, state keys, remove(), reparameterization_to_host()."""
    from hcp_diffusion_amd.layers import HipLinear
    dev = backend.device
    parent = torch.nn.Module(); parent.fc = HipLinear(64, 40).to(dev)
    parent.requires_grad_(False)
    blocks = LoraHipLayer.wrap_model(0, parent.fc, parent_block=parent, host_name="fc", rank=4, alpha=2.0)
    blk = blocks[""]
    assert type(parent.fc).__name__ == "LoraHipContainer" and blk.name == "lora_block_0"
    assert sorted(parent.state_dict().keys()) == ["fc._host.bias", "fc._host.weight", "fc.lora_block_0.alpha",
                                                   "fc.lora_block_0.layer.W_down", "fc.lora_block_0.layer.W_up"]
    assert abs(float(blk.alpha) - 0.5) < 1e-7 and blk.layer.W_up.abs().max().item() == 0
    with torch.no_grad():
        blk.layer.W_up.normal_(0, 0.1)
    x = torch.randn(3, 5, 64).to(torch.bfloat16)
    y = parent.fc(x.to(dev)).float().cpu()
    w_eff = parent.fc._host.weight.cpu() + float(blk.alpha) * (blk.layer.W_up.cpu() @ blk.layer.W_down.cpu())
    ref = x.float() @ w_eff.T + parent.fc._host.bias.cpu()
    assert ((y - ref).abs().max() / ref.abs().max()).item() < 2e-2
    blk.reparameterization_to_host()
    assert torch.allclose(parent.fc._host.weight.cpu(), w_eff, atol=1e-5)
    blk.remove()
    assert isinstance(parent.fc, HipLinear)


@pytest.mark.gpu
def test_sd15_full_size_forward_and_lora_grads_vs_golden():
    """Full SD1.5 architecture (859.5 M params, seeded init), batch 1, 64x64 latents, 77x768 context, LoRA rank 8.
    The fp32 oracle ran in the build container (oracle/make_golden.py -> tests/golden/sd15_full_oracle.pt: full prediction,
    loss, and a per-tensor fingerprint (norm, seeded projection) of all 320 LoRA gradient tensors)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import grad_fingerprint, sd15_full_inputs, sd15_lora_init_
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sd15_full_oracle.pt"))
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel()
    nat = seeded_init_(nat.to_empty(device=dev), 1)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=8)], lr=1e-4)
    assert len(tr.bucket.blocks) == g["n_lora"] == 160 and tr.bucket.numel == 2_992_128   # 160 layers: 224*sum(C)+16*12288
    lora_named = [(n, p) for n, p in nat.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    tr.bucket.pack()
    x0, ehs, noise, t = sd15_full_inputs()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev)).sample.cpu()
    assert ((pred - g["pred"]).norm() / g["pred"].norm()).item() < 2e-2
    loss = tr.forward_backward(x0.to(dev), ehs.to(dev)).item()
    assert abs(loss - g["loss"]) / g["loss"] < 2e-2
    fp = grad_fingerprint([(n, p.grad) for n, p in lora_named])
    import math
    num = sum(fp[n][1] * g["fingerprint"][n][1] for n in fp); da = math.sqrt(sum(v[1] ** 2 for v in fp.values()))
    db = math.sqrt(sum(v[1] ** 2 for v in g["fingerprint"].values()))
    assert num / (da * db) > 0.99                                # projections agree in sign and size across 320 tensors
    bad = [n for n in fp if g["fingerprint"][n][0] > 1e-7 and abs(fp[n][0] - g["fingerprint"][n][0]) / g["fingerprint"][n][0] > 0.1]
    assert len(bad) <= 3, bad                                    # per-tensor gradient norms within 10% (bf16 pipeline)


@pytest.mark.gpu
def test_sd15_full_size_batch4_blocks_and_full_lora_gradient_vs_golden():
    """The BENCHMARK shape (BASELINE.json configs[1]: SD1.5, B=4, 64x64 latents, LoRA r=8, timesteps 10/250/500/999) against the fp32
    oracle (tests/golden/sd15_full_b4_oracle.pt, oracle/make_golden.sd15_full_b4_vectors), at the tolerances SURVEY.md §8(c) states:
    every block-boundary activation (cumulative from the input; seeded 8192-element samples) rel-L2 <= 1.5e-2, prediction rel-L2 <= 2e-2, loss <= 1e-2 relative,
    cosine of the FULL flat LoRA gradient (2,992,128 elements) >= 0.999."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import SD15_BOUNDARIES, boundary_sample, dequantize_grads, sd15_b4_inputs, sd15_lora_init_
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sd15_full_b4_oracle.pt"))
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel()
    nat = seeded_init_(nat.to_empty(device=dev), 1)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=8)], lr=1e-4)
    by_name = {n: p for n, p in nat.named_parameters() if "lora_block_" in n}
    assert sorted(by_name) == sorted(g["grad_names"])
    lora_named = [(n, by_name[n]) for n in g["grad_names"]]             # the oracle's enumeration order (the flat gradient's layout)
    sd15_lora_init_(lora_named)
    tr.bucket.pack()
    x0, ehs, noise, t = sd15_b4_inputs()
    named = dict(nat.named_modules())
    got, hooks = {}, []
    for name in SD15_BOUNDARIES:
        def hook(mod, args, out, name=name):
            got[name] = out[0] if isinstance(out, tuple) else out
        hooks.append(named[name].register_forward_hook(hook))
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev)).sample.cpu()
    for h in hooks:
        h.remove()
    worst = {}
    for name in SD15_BOUNDARIES:
        if name == "conv_norm_out":                    # the native module returns GroupNorm+SiLU fused (one kernel): compare after SiLU
            ref, ref_n = g["boundaries"][name]
            smp, _ = boundary_sample(name, got[name].permute(0, 3, 1, 2).float().contiguous())
            worst[name] = ((smp - F.silu(ref)).norm() / F.silu(ref).norm()).item()
            continue
        y = got[name].permute(0, 3, 1, 2)              # native activations are [B,H,W,C]; the samples are taken at logical NCHW positions
        smp, nrm = boundary_sample(name, y.float().contiguous())
        ref, ref_n = g["boundaries"][name]
        worst[name] = ((smp - ref).norm() / ref.norm()).item()
        assert abs(nrm - ref_n) / ref_n < 1e-2, (name, nrm, ref_n)
    # cumulative error from the input through every preceding block (measured on MI355X: 0.003 after conv_in, 0.008 after
    # down_blocks.0, 0.010-0.014 from down_blocks.1 on): below the end-to-end bound everywhere, and below SURVEY's per-block 1e-2
    # for the first blocks, where 'per block' and 'cumulative' still coincide
    assert max(worst.values()) < 1.5e-2 and worst["conv_in"] < 5e-3 and worst["down_blocks.0"] < 1e-2, worst
    assert ((pred - g["pred"]).norm() / g["pred"].norm()).item() < 2e-2
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    loss = tr.forward_backward(x0.to(dev), ehs.to(dev)).item()
    assert abs(loss - g["loss"]) / g["loss"] < 1e-2
    flat = torch.cat([p.grad.detach().float().flatten().cpu() for _, p in lora_named])
    ref = dequantize_grads(g["grad_q"], g["grad_scales"], lora_named)
    cos = (flat.double() @ ref.double() / (flat.double().norm() * ref.double().norm())).item()      # fp64: 3M-term dot product
    print(f"[b4] worst block rel-L2 {max(worst.values()):.2e}, LoRA gradient cosine {cos:.5f}, norm {flat.norm().item():.5f} vs {g['grad_norm']:.5f}")
    assert cos > 0.999 and abs(flat.norm().item() - g["grad_norm"]) / g["grad_norm"] < 2e-2


def test_fp16_mixed_precision_is_refused_loudly(backend):
    """The reference's DEFAULT mixed_precision is 'fp16' (cfgs/train/train_base.yaml:2, train_ac.py:116-123); the native path is bf16-only
    and says so at the first UNet call instead of silently computing in another precision (Precision)."""
    _, nat = _pair(TINY_CONFIG, backend.device)
    g = torch.load(os.path.join(GOLD, "tiny_unet_oracle.pt"))
    with pytest.raises(NotImplementedError, match="mixed_precision 'fp16'"):
        nat(backend.to(g["x0"].half()), backend.to(g["t"]), backend.to(g["ehs"]))
    with pytest.raises(NotImplementedError, match="mixed_precision 'fp16'"):
        nat(backend.to(g["x0"]), backend.to(g["t"]), backend.to(g["ehs"].half()))
    if backend.is_gpu:
        with torch.autocast("cuda", dtype=torch.float16), pytest.raises(NotImplementedError, match="mixed_precision 'fp16'"):
            nat(backend.to(g["x0"]), backend.to(g["t"]), backend.to(g["ehs"]))
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():            # the supported mode is untouched
            assert torch.isfinite(nat(backend.to(g["x0"]), backend.to(g["t"]), backend.to(g["ehs"])).sample).all()


def _per_block_errors(ora, nat, to_dev, xt, t, ehs, autocast_too=False, **fwd_kw):
    """TRUE per-block errors (SURVEY.md §8c "rel-L2 <= 1e-2 per block"): every down / mid / up block of the NATIVE model is fed the
    ORACLE's input of that block — hidden state and, for the up blocks, the skip tensors (forward pre-hooks swap them in during ONE
    native forward; the time embedding and prompt states stay the native model's own: they are not block outputs) — and its output
    is compared with the oracle block's output on the same input.  Nothing accumulates from block to block."""
    names = ([f"down_blocks.{i}" for i in range(len(ora.down_blocks))] + ["mid_block"] + [f"up_blocks.{i}" for i in range(len(ora.up_blocks))])
    o_mod, n_mod = dict(ora.named_modules()), dict(nat.named_modules())
    rec, got, hooks = {}, {}, []
    for nm in names:
        hooks.append(o_mod[nm].register_forward_hook(lambda m, a, out, nm=nm: rec.__setitem__(nm, (a, out))))
    with torch.no_grad():
        ora(xt, t, ehs, **fwd_kw)
    for h in hooks:
        h.remove()
    nhwc = lambda y: to_dev(y.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
    hooks = []
    for nm in names:
        def pre(mod, args, nm=nm):
            a = rec[nm][0]
            if nm.startswith("up_blocks"):                       # (h, skips, temb, ctx)
                return (nhwc(a[0]), tuple(nhwc(s) for s in a[1])) + tuple(args[2:])
            return (nhwc(a[0]),) + tuple(args[1:])               # (h, temb, ctx)
        hooks.append(n_mod[nm].register_forward_pre_hook(pre))
        hooks.append(n_mod[nm].register_forward_hook(lambda m, a, out, nm=nm: got.__setitem__(nm, out[0] if isinstance(out, tuple) else out)))
    with torch.no_grad():
        nat(to_dev(xt), to_dev(t), to_dev(ehs), **{k: ({kk: to_dev(vv) for kk, vv in v.items()} if isinstance(v, dict) else to_dev(v)) for k, v in fwd_kw.items()})
    for h in hooks:
        h.remove()
    errs, errs_ac = {}, {}
    b16 = lambda v: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else
                     tuple(b16(u) for u in v) if isinstance(v, tuple) else v)
    for nm in names:
        ref = rec[nm][1][0] if isinstance(rec[nm][1], tuple) else rec[nm][1]
        y = got[nm].permute(0, 3, 1, 2).float().cpu()
        errs[nm] = ((y - ref).norm() / ref.norm()).item()
        if autocast_too:            # the same oracle block under torch.autocast(bfloat16) — the reference's execution mode, train_ac.py:449 — on the same input
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                ya = o_mod[nm](*[b16(v) for v in rec[nm][0]])
            ya = (ya[0] if isinstance(ya, tuple) else ya).float()
            errs_ac[nm] = ((ya - ref).norm() / ref.norm()).item()
    return (errs, errs_ac) if autocast_too else errs


def test_per_block_error_tiny(backend):
    """The per-block harness on the interpreter (TINY config): oracle block input -> native block, <= 1e-2 per block."""
    ora, nat = _pair(TINY_CONFIG, backend.device)
    g = torch.load(os.path.join(GOLD, "tiny_unet_oracle.pt"))
    xt = add_noise(g["x0"], g["noise"], g["t"], ddpm_alphas_cumprod())
    errs, errs_ac = _per_block_errors(ora, nat, backend.to, xt, g["t"], g["ehs"], autocast_too=True)
    assert len(errs) == len(ora.down_blocks) + 1 + len(ora.up_blocks) and max(errs.values()) < 1e-2, errs
    # ... and no block is less precise than the same oracle block under bf16 autocast (the reference's execution mode) by more than 10 %
    assert all(errs[k] < 1.1 * errs_ac[k] for k in errs), (errs, errs_ac)


@pytest.mark.gpu
def test_sd15_full_size_batch4_true_per_block_error():
    """SURVEY.md §8(c) says rel-L2 <= 1e-2 PER BLOCK; the golden-fixture test above can only bound the CUMULATIVE error
    (its fixtures hold samples of the oracle's boundaries, not the 21 MB tensors a block would need as input).  Here the fp32 oracle
    itself runs on the host cores (full SD1.5, B = 4, 64x64 latents, the benchmark shape, timesteps 10/250/500/999) and every native
    block gets the oracle's input of that block: nine blocks, each <= 1e-2 on its own."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.make_golden import sd15_b4_inputs
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    ora = seeded_init_(OracleUNet2DConditionModel(), 1)
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel()
    nat = seeded_init_(nat.to_empty(device=dev), 1)
    x0, ehs, noise, t = sd15_b4_inputs()
    xt = add_noise(x0, noise, t, ddpm_alphas_cumprod())
    errs, errs_ac = _per_block_errors(ora, nat, lambda v: v.to(dev), xt, t, ehs, autocast_too=True)
    print("[b4 per-block rel-L2, native (the same oracle block under bf16 autocast)] " + ", ".join(f"{k} {v:.2e} ({errs_ac[k]:.2e})" for k, v in errs.items()))
    assert len(errs) == 9 and max(errs.values()) < 1e-2, errs
    # round 5: block by block the native arithmetic is at least as precise as the reference's bf16 execution mode
    # (measured 0.90-0.99 of the autocast error; tools/diag/sdxl_block_diag.py does the same per module for SDXL)
    assert all(errs[k] < 1.1 * errs_ac[k] for k in errs), (errs, errs_ac)


@pytest.mark.gpu
def test_sdxl_full_size_forward_and_lora_grads_vs_golden():
    """Full SDXL-base architecture (2.567 B params, seeded init), batch 1, 64x64 latents, 77x2048 context + text_time
    conditioning, LoRA rank 16 (BASELINE.json configs[3] layer shapes).  Oracle values: oracle/make_golden.py sdxl ->
    tests/golden/sdxl_full_oracle.pt (generated in the build container)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import math
    from oracle.make_golden import grad_fingerprint, sd15_lora_init_, sdxl_full_inputs
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    g = torch.load(os.path.join(GOLD, "sdxl_full_oracle.pt"))
    with torch.device("meta"):
        nat = NativeUNet2DConditionModel(**SDXL_CONFIG)
    nat = seeded_init_(nat.to_empty(device=dev), 1)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=16)], lr=1e-4)
    assert len(tr.bucket.blocks) == g["n_lora"] and tr.bucket.numel == g["n_lora_params"]
    lora_named = [(n, p) for n, p in nat.named_parameters() if "lora_block_" in n]
    sd15_lora_init_(lora_named)
    tr.bucket.pack()
    x0, ehs, noise, t, added = sdxl_full_inputs()
    added = {k: v.to(dev) for k, v in added.items()}
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    with torch.no_grad():
        pred = nat(K.add_noise(x0.to(dev), noise.to(dev), t.to(dev), tr.acp), t.to(dev), ehs.to(dev), added_cond_kwargs=added).sample.cpu()
    assert ((pred - g["pred"]).norm() / g["pred"].norm()).item() < 3e-2         # 70 transformer blocks deep in bf16
    loss = tr.forward_backward(x0.to(dev), ehs.to(dev), None, added).item()
    assert abs(loss - g["loss"]) / g["loss"] < 2e-2
    fp = grad_fingerprint([(n, p.grad) for n, p in lora_named])
    num = sum(fp[n][1] * g["fingerprint"][n][1] for n in fp); da = math.sqrt(sum(v[1] ** 2 for v in fp.values()))
    db = math.sqrt(sum(v[1] ** 2 for v in g["fingerprint"].values()))
    assert num / (da * db) > 0.99
    bad = [n for n in fp if g["fingerprint"][n][0] > 1e-7 and abs(fp[n][0] - g["fingerprint"][n][0]) / g["fingerprint"][n][0] > 0.1]
    assert len(bad) <= len(fp) // 100, bad


@pytest.mark.parametrize("host", ["sd15", "sdxl"])
def test_tiny_controlnet_train_step_vs_oracle(backend, host):
    """ControlNet branch (reference hcpdiff/models/controlnet.py, cfgs/plugins/plugin_controlnet.yaml): frozen host UNet,
    trainable deep copy of its encoder + cond_head + zero convs, wired in through the reference's hook layout; all branch
    gradients vs fp32 autograd of the oracle restatement, then clip + AdamW.  Zero convs get non-zero seeded values (a branch
    "after some training"), otherwise every gradient upstream of them is exactly zero.  host = 'sdxl': a text_time host (three
    down blocks, no attention in the first, added_cond_kwargs on the host call) — the branch is copied the same way and, like the
    reference's (controlnet.py:19-25,88-97), carries no text_time term in its own time embedding."""
    from hcp_diffusion_amd.controlnet import make_controlnet
    from oracle.unet_sd15 import OracleControlNet
    dev = backend.device
    from oracle.unet_sd15 import TINY_SDXL_CONFIG
    cfg = TINY_SDXL_CONFIG if host == "sdxl" else TINY_CONFIG if backend.is_gpu else MICRO_CONFIG          # the interpreter gets the two-level miniature
    ora, nat = _pair(cfg, dev)
    ora.requires_grad_(False)
    torch.manual_seed(3)
    ocn = OracleControlNet(ora)
    for n_, p_ in ocn.named_parameters():
        p_.requires_grad_(True)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n_, p_ in ocn.named_parameters():
            if n_.startswith(("cond_head", "controlnet_")):
                p_.copy_(torch.randn(p_.shape, generator=g) * (0.3 / max(1.0, p_[0].numel() ** 0.5) if p_.dim() > 1 else 0.05))
    plug = make_controlnet(nat)
    assert sorted(k for k, _ in plug.named_parameters()) == sorted(k for k, _ in ocn.named_parameters())
    assert plug.cond_head[0].weight.shape == (16, 3, 3, 3) and [m.stride[0] for m in plug.cond_head if hasattr(m, "stride")] == [1, 1, 2, 1, 2, 1, 2, 1]
    cd = cfg["cross_attention_dim"]
    plug.load_state_dict(ocn.state_dict())
    tr = NativeTrainer(nat, None, lr=1e-3, plugins=[(plug, 1e-3)])
    assert not any(p.requires_grad for p in nat.parameters()) and all(p.requires_grad for p in plug.parameters())
    g2 = torch.Generator().manual_seed(21)
    B = 2
    x0 = torch.randn(B, 4, 8, 8, generator=g2); ehs = torch.randn(B, 24, cd, generator=g2)
    noise = torch.randn(B, 4, 8, 8, generator=g2); t = torch.tensor([30, 800]); cond = torch.rand(B, 3, 64, 64, generator=g2)
    xt = add_noise(x0, noise, t, ddpm_alphas_cumprod())
    added = None
    if host == "sdxl":
        added = dict(text_embeds=torch.randn(B, cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"], generator=g2),
                     time_ids=torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]] * B))
    okw = dict(added_cond_kwargs=added) if added else {}
    pred = ora(xt, t, ehs, control_residuals=ocn(xt, t, ehs, cond), **okw).sample
    loss_o = F.mse_loss(pred, noise)
    loss_o.backward()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    nadded = {k: v.to(dev) for k, v in added.items()} if added else None
    loss_n = tr.forward_backward(x0.to(dev), ehs.to(dev), None, nadded, dict(cond=cond.to(dev)))
    assert abs(loss_o.item() - loss_n.item()) / loss_o.item() < 2e-2
    po = dict(ocn.named_parameters())
    num = da = db = 0.0
    bad = []
    for name, p in plug.named_parameters():
        go, gn = po[name].grad, p.grad.cpu()
        num += (go * gn).sum().item(); da += go.norm().item() ** 2; db += gn.norm().item() ** 2
        cos = F.cosine_similarity(go.flatten(), gn.flatten(), dim=0).item()
        if go.norm().item() > 1e-7 and cos < 0.97:
            bad.append((name, round(cos, 4)))
    assert num / (da * db) ** 0.5 > 0.995 and not bad, bad
    tr.all_reduce(); tr.optimizer_step()
    assert tr.host_buckets[0].bucket.grads.abs().max().item() == 0.0
    plug.remove()
    with torch.no_grad():                                              # hooks gone: the host is the plain UNet again
        nkw = dict(added_cond_kwargs=nadded) if added else {}
        y0 = nat(backend.to(xt), backend.to(t), backend.to(ehs), **nkw).sample.cpu()
        assert ((y0 - ora(xt, t, ehs, **okw).sample).norm() / y0.norm()).item() < 2e-2


def test_two_dataset_step_accumulates_like_reference(backend):
    """train_ac.py:467-504: one batch per dataset, every backward accumulates, ONE optimizer step.  Two half batches with
    loss weights (1, 0.5) must give the gradient of loss_a + 0.5 * loss_b."""
    dev = backend.device
    _, nat = _pair(MICRO_CONFIG, dev)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-3)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in tr.bucket.blocks:
            blk.layer.W_up.copy_(backend.to(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05))
    tr.bucket.pack()
    x = [backend.to(torch.randn(1, 4, 8, 8, generator=g)) for _ in range(2)]
    e = [backend.to(torch.randn(1, 24, 32, generator=g)) for _ in range(2)]
    n = [backend.to(torch.randn(1, 4, 8, 8, generator=g)) for _ in range(2)]
    t = [backend.to(torch.tensor([100])), backend.to(torch.tensor([650]))]
    cur = {"i": 0}
    tr.make_noise = lambda lat: (K.add_noise(lat, n[cur["i"]], t[cur["i"]], tr.acp), n[cur["i"]], t[cur["i"]])
    grads = []
    for i in range(2):
        cur["i"] = i
        tr.forward_backward(x[i], e[i])
        grads.append(tr.bucket.grads.clone()); tr.bucket.grads.zero_()
    expect = grads[0] + 0.5 * grads[1]
    seq = iter([0, 1])
    tr.make_noise = lambda lat: (lambda i: (K.add_noise(lat, n[i], t[i], tr.acp), n[i], t[i]))(next(seq))
    tr.optimizer_step = lambda: None                              # look at the accumulated gradient before it is consumed
    loss = tr.train_data_list([dict(latents=x[0], encoder_hidden_states=e[0]),
                               dict(latents=x[1], encoder_hidden_states=e[1], loss_weight=0.5)])
    assert loss.numel() == 1
    assert ((tr.bucket.grads - expect).norm() / expect.norm()).item() < 1e-5


@pytest.mark.parametrize("tag", ["conv3x3_r4", "conv3x3_s2_r8"])
def test_conv_lora_layer_vs_reference_golden(backend, tag):
    """LoCon: native conv LoRA (side path T = conv3x3(x, W_down), K-extension with alpha W_up) against vectors produced by the
    reference's own LoraLayer.Conv2dLayer (tests/golden/lora_reference.pt, oracle/make_golden.py)."""
    from hcp_diffusion_amd.layers import HipConv2d
    g = torch.load(os.path.join(GOLD, "lora_reference.pt"))[tag]
    dev = backend.device
    cout, cin = g["host_weight"].shape[:2]
    parent = torch.nn.Module(); parent.conv = HipConv2d(cin, cout, 3, g["stride"], 1)
    with torch.no_grad():
        parent.conv.weight.copy_(g["host_weight"]); parent.conv.bias.copy_(g["host_bias"])
    parent.to(dev).requires_grad_(False)
    blk = LoraHipLayer.wrap_model(0, parent.conv, parent_block=parent, host_name="conv", rank=g["rank"], alpha=g["cfg_alpha"])[""]
    assert sorted(parent.state_dict().keys()) == g["state_keys"] and tuple(blk.layer.W_down.shape) == tuple(g["W_down"].shape)
    assert torch.equal(blk.alpha.cpu(), g["alpha_buffer"])
    with torch.no_grad():
        blk.layer.W_down.copy_(backend.to(g["W_down"])); blk.layer.W_up.copy_(backend.to(g["W_up"]))
    blk.packed()                                          # stand-alone layer: creates its private one-layer bucket
    x = backend.to(g["x"].permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)).requires_grad_(True)
    y = parent.conv(x)
    y.backward(backend.to(g["dy"].permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)))
    def rel(a, b):
        return ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach().permute(0, 3, 1, 2), g["y"]) < 2e-2 and rel(x.grad.permute(0, 3, 1, 2), g["dx"]) < 2e-2
    assert rel(blk.layer.W_down.grad, g["dW_down"]) < 2e-2 and rel(blk.layer.W_up.grad, g["dW_up"]) < 2e-2
    w_eff = g["host_weight"] + float(blk.alpha) * torch.einsum("or,rikl->oikl", g["W_up"][:, :, 0, 0], g["W_down"])
    blk.reparameterization_to_host()
    assert torch.allclose(parent.conv._host.weight.cpu(), w_eff, atol=1e-5)


@pytest.mark.parametrize("edge_convs", [False, True])
def test_tiny_locon_train_step_vs_oracle(backend, edge_convs):
    """cfgs/train/examples/locon.yaml: LoRA on the attention / ff Linear layers AND on the resnets' convs, proj_in/out 1x1 convs
    and the down/upsampler convs.  edge_convs: conv_in / conv_out as well (4 latent channels: the merged-weight form, in the captured
    trainer step like every other layer)."""
    pats_conv = [r"re:.*\.resnets$", r"re:.*\.proj_in$", r"re:.*\.proj_out$", r"re:.*\.conv$"] + ([r"re:conv_in$", r"re:conv_out$"] if edge_convs else [])
    dev = backend.device
    ora, nat = _pair(TINY_CONFIG, dev)
    ora.requires_grad_(False)
    wr = wrap_lora(ora, PATS, rank=4)
    wr.update(wrap_lora(ora, pats_conv, rank=4, conv=True))
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4), dict(layers=pats_conv, rank=4)], lr=1e-3)
    norm = lambda k: k.replace("lora_block_1", "lora_block_0")     # the reference names blocks by cfg-list index; the oracle always 0
    assert sorted(k for k in ora.state_dict() if "lora" in k) == sorted(norm(k) for k in nat.state_dict() if "lora" in k)
    assert {tuple(v.shape) for k, v in nat.state_dict().items() if k.endswith("conv1.lora_block_1.layer.W_down")} >= {(4, 80, 3, 3)}
    assert any(k.endswith("proj_in.lora_block_1.layer.W_down") and v.dim() == 4 for k, v in nat.state_dict().items())
    gen = torch.Generator().manual_seed(5)
    sd_n = dict(nat.named_parameters())
    with torch.no_grad():
        for name, p in ora.named_parameters():
            if "lora_block_" in name:
                if name.endswith("W_up"):
                    p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
                sd_n[name.replace("lora_block_0", "lora_block_0" if name in sd_n else "lora_block_1")].copy_(p)
    tr.bucket.pack()
    g2 = torch.Generator().manual_seed(42)
    x0 = torch.randn(2, 4, 8, 8, generator=g2); ehs = torch.randn(2, 77, 64, generator=g2)
    noise = torch.randn(2, 4, 8, 8, generator=g2); t = torch.tensor([40, 710])
    pred = ora(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs).sample
    loss_o = F.mse_loss(pred, noise)
    loss_o.backward()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    loss_n = tr.forward_backward(x0.to(dev), ehs.to(dev))
    assert abs(loss_o.item() - loss_n.item()) / loss_o.item() < 2e-2
    num = da = db = 0.0
    bad = []
    for name, p in ora.named_parameters():
        if "lora_block_" not in name:
            continue
        gn = sd_n[name if name in sd_n else name.replace("lora_block_0", "lora_block_1")].grad.cpu()
        go = p.grad
        num += (go * gn).sum().item(); da += go.norm().item() ** 2; db += gn.norm().item() ** 2
        if go.norm().item() > 1e-7 and F.cosine_similarity(go.flatten(), gn.flatten(), dim=0).item() < 0.97:
            bad.append(name)
    assert num / (da * db) ** 0.5 > 0.995 and not bad, bad


def test_dapp_layer_positive_and_negative_branches(backend):
    """`lora_layer_map['dapp_hip']` (VERDICT r4 missing #3): the reference's DAPPPatchContainer (lora_layers_patch.py:102-135) runs the
    first batch half with the summed 'n'-branch blocks and the second half with the 'p'-branch blocks.  Two p blocks + one n block on one
    Linear host, with a fused residual, against fp32 autograd on the two merged weights."""
    from hcp_diffusion_amd.layers import HipLinear
    from hcp_diffusion_amd.lora import DAPPHipLayer, lora_layer_map
    assert lora_layer_map["dapp_hip"] is DAPPHipLayer
    dev = backend.device
    torch.manual_seed(8)
    parent = torch.nn.Module(); parent.fc = HipLinear(64, 48).to(dev)
    parent.requires_grad_(False)
    bp0 = DAPPHipLayer.wrap_model(0, parent.fc, parent_block=parent, host_name="fc", rank=4, alpha=1.0, branch="p")[""]
    bn = DAPPHipLayer.wrap_model(1, parent.fc, parent_block=parent, host_name="fc", rank=8, alpha=2.0, branch="n")[""]
    bp1 = DAPPHipLayer.wrap_model(2, parent.fc, parent_block=parent, host_name="fc", rank=4, alpha=3.0, branch="p")[""]
    assert type(parent.fc).__name__ == "DAPPHipContainer" and parent.fc.plugin_names == ["lora_block_0", "lora_block_1", "lora_block_2"]
    with torch.no_grad():
        for b in (bp0, bn, bp1):
            b.layer.W_up.normal_(0, 0.1)
    x = torch.randn(6, 7, 64).to(torch.bfloat16); res = torch.randn(6, 7, 48).to(torch.bfloat16); dy = torch.randn(6, 7, 48).to(torch.bfloat16)
    fac = {b: (b.layer.W_down.detach().cpu().clone().requires_grad_(True), b.layer.W_up.detach().cpu().clone().requires_grad_(True)) for b in (bp0, bn, bp1)}
    host = parent.fc._host
    w_of = lambda bs: host.weight.cpu() + sum(float(b.alpha) * (fac[b][1] @ fac[b][0]) for b in bs)
    xr = x.float().requires_grad_(True)
    yr = torch.cat([xr[:3] @ w_of([bn]).T, xr[3:] @ w_of([bp0, bp1]).T]) + host.bias.cpu() + res.float()
    yr.backward(dy.float())
    xn = backend.to(x).requires_grad_(True)
    y = parent.fc(xn, residual=backend.to(res))
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach(), yr.detach()) < 2e-2 and rel(xn.grad, xr.grad) < 2e-2
    for b in (bp0, bn, bp1):
        assert rel(b.layer.W_down.grad, fac[b][0].grad) < 3e-2 and rel(b.layer.W_up.grad, fac[b][1].grad) < 3e-2
    # the LAST plugin's dropout acts on BOTH halves (lora_layers_patch.py:132-133: `self[name].post_forward` after the loop), whichever
    # branch that plugin belongs to: with p = 1 on the last ('p') block the 'n' half is dropped too and only the fused residual remains
    bp1.dropout.p = 1.0
    parent.fc.train()
    y1 = parent.fc(backend.to(x), residual=backend.to(res))
    assert torch.equal(y1.detach().cpu(), res)
    bp1.dropout.p = 0.0
    # per-sample keyword arguments of a 3x3 conv host follow their half of the batch
    from hcp_diffusion_amd.layers import HipConv2d
    cpar = torch.nn.Module(); cpar.cv = HipConv2d(64, 64, 3, padding=1).to(dev)
    cpar.requires_grad_(False)
    cblk = [DAPPHipLayer.wrap_model(i, cpar.cv, parent_block=cpar, host_name="cv", rank=4, alpha=1.0, branch=br)[""] for i, br in enumerate("pn")]
    with torch.no_grad():
        for b in cblk:
            b.layer.W_up.normal_(0, 0.1)
    xc = torch.randn(4, 8, 8, 64).to(torch.bfloat16); rb = torch.randn(4, 64)
    yc = cpar.cv(backend.to(xc), rowbias=backend.to(rb))
    chost = cpar.cv._host
    wc = lambda b: chost.weight.detach().float().cpu() + float(b.alpha) * torch.einsum(
        "or,rikl->oikl", b.layer.W_up.detach().float().cpu()[:, :, 0, 0], b.layer.W_down.detach().float().cpu())
    xcr = xc.float().permute(0, 3, 1, 2)
    ycr = torch.cat([torch.nn.functional.conv2d(xcr[:2], wc(cblk[1]), chost.bias.detach().float().cpu(), padding=1),
                     torch.nn.functional.conv2d(xcr[2:], wc(cblk[0]), chost.bias.detach().float().cpu(), padding=1)]) + rb[:, :, None, None]
    assert rel(yc.detach().permute(0, 3, 1, 2), ycr) < 2e-2
    # one branch only: refused like the reference (which adds None to the host weight)
    solo = torch.nn.Module(); solo.fc = HipLinear(64, 48).to(dev)
    DAPPHipLayer.wrap_model(0, solo.fc, parent_block=solo, host_name="fc", rank=4, branch="p")
    with pytest.raises(ValueError, match="at least one 'p' and one 'n'"):
        solo.fc(backend.to(x))


def test_two_lora_blocks_on_one_host(backend):
    """Two cfg groups matching the same layer -> lora_block_0 and lora_block_1 on one container; the reference sums their
    get_weight() (lora_base_patch.py:20-27).  Native: adjacent rank slots of one fused-LoRA GEMM."""
    from hcp_diffusion_amd.layers import HipLinear
    dev = backend.device
    torch.manual_seed(6)
    parent = torch.nn.Module(); parent.fc = HipLinear(64, 48).to(dev)
    parent.requires_grad_(False)
    b0 = LoraHipLayer.wrap_model(0, parent.fc, parent_block=parent, host_name="fc", rank=4, alpha=1.0)[""]
    b1 = LoraHipLayer.wrap_model(1, parent.fc, parent_block=parent, host_name="fc", rank=8, alpha=4.0)[""]
    assert type(parent.fc).__name__ == "LoraHipContainer" and parent.fc.plugin_names == ["lora_block_0", "lora_block_1"]
    with torch.no_grad():
        b0.layer.W_up.normal_(0, 0.1); b1.layer.W_up.normal_(0, 0.1)
    x = torch.randn(5, 7, 64).to(torch.bfloat16)
    dy = torch.randn(5, 7, 48).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    wd0, wu0, wd1, wu1 = (t.detach().cpu().clone().requires_grad_(True) for t in (b0.layer.W_down, b0.layer.W_up, b1.layer.W_down, b1.layer.W_up))
    w_eff = parent.fc._host.weight.cpu() + float(b0.alpha) * (wu0 @ wd0) + float(b1.alpha) * (wu1 @ wd1)
    yr = xr @ w_eff.T + parent.fc._host.bias.cpu()
    yr.backward(dy.float())
    xn = backend.to(x).requires_grad_(True)
    y = parent.fc(xn)
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach(), yr.detach()) < 2e-2 and rel(xn.grad, xr.grad) < 2e-2
    for blk, gd, gu in ((b0, wd0.grad, wu0.grad), (b1, wd1.grad, wu1.grad)):
        assert rel(blk.layer.W_down.grad, gd) < 2e-2 and rel(blk.layer.W_up.grad, gu) < 2e-2
    b1.remove()
    assert parent.fc.plugin_names == ["lora_block_0"]
    y1 = parent.fc(backend.to(x)).float().cpu()
    ref1 = x.float() @ (parent.fc._host.weight.cpu() + float(b0.alpha) * (wu0 @ wd0)).T.detach() + parent.fc._host.bias.cpu()
    assert rel(y1, ref1.detach()) < 2e-2


@pytest.mark.parametrize("ranks", [(32, 32), (40, 8), (16, 12, 8)])
def test_lora_blocks_on_one_host_beyond_32_slots(backend, ranks):
    """Stacked LoRA blocks whose ranks no longer fit one 32-slot group (two rank-32 LoRAs on one layer; a wide block among them): the
    reference still just sums get_weight() (lora_base_patch.py:20-27); natively the blocks share the wide form — one skinny side GEMM over
    all slots, K-extension of ceil(slots / 32) * 32 columns — and each block's gradients are cut out by slot offset."""
    from hcp_diffusion_amd.layers import HipLinear
    dev = backend.device
    torch.manual_seed(sum(ranks))
    parent = torch.nn.Module(); parent.fc = HipLinear(72, 40).to(dev)
    parent.requires_grad_(False)
    blks = [LoraHipLayer.wrap_model(i, parent.fc, parent_block=parent, host_name="fc", rank=r, alpha=float(2 + i))[""] for i, r in enumerate(ranks)]
    assert parent.fc.plugin_names == [f"lora_block_{i}" for i in range(len(ranks))]
    with torch.no_grad():
        for b in blks:
            b.layer.W_up.normal_(0, 0.1)
    x = torch.randn(6, 9, 72).to(torch.bfloat16); dy = torch.randn(6, 9, 40).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    fac = [(b.layer.W_down.detach().cpu().clone().requires_grad_(True), b.layer.W_up.detach().cpu().clone().requires_grad_(True)) for b in blks]
    w_eff = parent.fc._host.weight.cpu() + sum(float(b.alpha) * (wu @ wd) for b, (wd, wu) in zip(blks, fac))
    yr = xr @ w_eff.T + parent.fc._host.bias.cpu()
    yr.backward(dy.float())
    xn = backend.to(x).requires_grad_(True)
    y = parent.fc(xn)
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach(), yr.detach()) < 2e-2 and rel(xn.grad, xr.grad) < 2e-2
    for b, (wd, wu) in zip(blks, fac):
        assert rel(b.layer.W_down.grad, wd.grad) < 2e-2 and rel(b.layer.W_up.grad, wu.grad) < 2e-2


@pytest.mark.parametrize("ranks", [(4, 8), (32, 8)])
@pytest.mark.parametrize("stride", [1, 2])
def test_two_lora_blocks_on_one_conv_host(backend, stride, ranks):
    """Two cfg groups matching the same 3x3 conv -> lora_block_0 and lora_block_1 on one container (the reference sums their
    get_weight(), lora_base_patch.py:20-27; LoCon factors lora_layers_patch.py:64-100).  Native: ONE skinny conv T = conv3x3(x, [W_down_0;
    W_down_1]) fills adjacent rank slots, the host conv takes T as its K-extension, and each block's gradients are cut out of the shared
    T / U by slot offset.  Checked against fp32 autograd on the merged weight."""
    from hcp_diffusion_amd.layers import HipConv2d
    dev = backend.device
    torch.manual_seed(8)
    cin, cout = 16, 24
    parent = torch.nn.Module(); parent.conv = HipConv2d(cin, cout, 3, stride, 1).to(dev)
    parent.requires_grad_(False)
    # ranks (32, 8): 40 rank slots do not fit the 32 of one side path -> the container switches to the merged-weight form
    b0 = LoraHipLayer.wrap_model(0, parent.conv, parent_block=parent, host_name="conv", rank=ranks[0], alpha=1.0)[""]
    b1 = LoraHipLayer.wrap_model(1, parent.conv, parent_block=parent, host_name="conv", rank=ranks[1], alpha=4.0)[""]
    assert type(parent.conv).__name__ == "LoraHipContainer" and parent.conv.plugin_names == ["lora_block_0", "lora_block_1"]
    assert tuple(b1.layer.W_down.shape) == (8, cin, 3, 3) and tuple(b1.layer.W_up.shape) == (cout, 8, 1, 1)
    with torch.no_grad():
        b0.layer.W_up.normal_(0, 0.1); b1.layer.W_up.normal_(0, 0.1)
    x = torch.randn(2, 6, 6, cin).to(torch.bfloat16)
    Ho = 6 // stride
    dy = torch.randn(2, Ho, Ho, cout).to(torch.bfloat16)
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wd0, wu0, wd1, wu1 = (t.detach().cpu().clone().requires_grad_(True) for t in (b0.layer.W_down, b0.layer.W_up, b1.layer.W_down, b1.layer.W_up))
    host = parent.conv._host
    w_eff = (host.weight.detach().cpu().float() + float(b0.alpha) * torch.einsum("or,rikl->oikl", wu0[:, :, 0, 0], wd0)
             + float(b1.alpha) * torch.einsum("or,rikl->oikl", wu1[:, :, 0, 0], wd1))
    yr = torch.nn.functional.conv2d(xr, w_eff, host.bias.detach().cpu().float(), stride=stride, padding=1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    xn = backend.to(x).requires_grad_(True)
    y = parent.conv(xn)
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < 2e-2 and rel(xn.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
    for blk, gd, gu in ((b0, wd0.grad, wu0.grad), (b1, wd1.grad, wu1.grad)):
        assert rel(blk.layer.W_down.grad, gd) < 2e-2 and rel(blk.layer.W_up.grad, gu) < 2e-2
    # the convolution's other operands (fused residual, per-sample row bias = the time embedding) through the same container call
    res = torch.randn(2, Ho, Ho, cout).to(torch.bfloat16)
    rb = torch.randn(2, cout)
    rr = res.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yk = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w_eff.detach(), host.bias.detach().cpu().float(), stride=stride, padding=1) \
        + rr + rb[:, :, None, None]
    yk.backward(dy.float().permute(0, 3, 1, 2))
    rn = backend.to(res).requires_grad_(True)
    y3 = parent.conv(backend.to(x), residual=rn, rowbias=backend.to(rb))
    y3.backward(backend.to(dy))
    assert rel(y3.detach().permute(0, 3, 1, 2), yk.detach()) < 2e-2 and rel(rn.grad.permute(0, 3, 1, 2), rr.grad) < 1e-2
    # a second step after the factors moved: the shared operand images follow both blocks
    for blk in (b0, b1):
        blk.layer.W_down.grad.zero_(); blk.layer.W_up.grad.zero_()
    with torch.no_grad():
        b0.layer.W_down.mul_(0.5); b1.layer.W_up.mul_(2.0)
    y2 = parent.conv(backend.to(x)).float().cpu()
    w2 = (host.weight.detach().cpu().float() + float(b0.alpha) * torch.einsum("or,rikl->oikl", wu0[:, :, 0, 0], 0.5 * wd0)
          + float(b1.alpha) * torch.einsum("or,rikl->oikl", 2.0 * wu1[:, :, 0, 0], wd1)).detach()
    ref2 = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w2, host.bias.detach().cpu().float(), stride=stride, padding=1)
    assert rel(y2.permute(0, 3, 1, 2), ref2) < 2e-2
    b1.remove()
    assert parent.conv.plugin_names == ["lora_block_0"]
    y1 = parent.conv(backend.to(x)).float().cpu()
    w1 = (host.weight.detach().cpu().float() + float(b0.alpha) * torch.einsum("or,rikl->oikl", wu0[:, :, 0, 0], 0.5 * wd0)).detach()
    ref1 = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w1, host.bias.detach().cpu().float(), stride=stride, padding=1)
    assert rel(y1.permute(0, 3, 1, 2), ref1) < 2e-2


def test_gradient_checkpointing_matches_plain_backward(backend):
    """model.gradient_checkpointing: True (reference default, train_base.yaml:69; wrapper.py:39-49): same loss and LoRA
    gradients as the un-checkpointed step (segments recomputed by the same kernels)."""
    dev = backend.device
    _, nat = _pair(MICRO_CONFIG, dev)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-3)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in tr.bucket.blocks:
            blk.layer.W_up.copy_(backend.to(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05))
    tr.bucket.pack()
    x0 = backend.to(torch.randn(2, 4, 8, 8, generator=g)); ehs = backend.to(torch.randn(2, 24, 32, generator=g))
    noise = backend.to(torch.randn(2, 4, 8, 8, generator=g)); t = backend.to(torch.tensor([12, 640]))
    tr.make_noise = lambda lat: (K.add_noise(lat, noise, t, tr.acp), noise, t)
    l0 = tr.forward_backward(x0, ehs).item()
    g0 = tr.bucket.grads.clone(); tr.bucket.grads.zero_()
    nat.enable_gradient_checkpointing()
    assert nat.down_blocks[0].gradient_checkpointing and nat.mid_block.gradient_checkpointing
    l1 = tr.forward_backward(x0, ehs).item()
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0))
    assert ((tr.bucket.grads - g0).norm() / g0.norm()).item() < 1e-4


def test_ema_matches_reference_schedule(backend):
    """model.ema (reference utils/ema.py ModelEMA): decay = clip(1 - (1 + step / inv_gamma)^-power, 0, decay_max), one fused
    launch per bucket with the step read on the device."""
    dev = backend.device
    _, nat = _pair(TINY_CONFIG, dev)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-2, ema=dict(decay_max=0.9997))
    ema_ref = tr.bucket.params.cpu().clone()
    g = torch.Generator().manual_seed(1)
    for step in (1, 2, 3):
        tr.bucket.grads.copy_(backend.to(torch.randn(tr.bucket.numel, generator=g)))
        tr.optimizer_step()
        decay = min(max(1 - (1 + step / 1.0) ** -(2 / 3), 0.0), 0.9997)
        ema_ref.lerp_(tr.bucket.params.cpu(), 1 - decay)
    assert ((tr._lora_state.ema.cpu() - ema_ref).abs().max() / ema_ref.abs().max()).item() < 1e-5
    sd = tr.ema_state_dict()
    name = next(n for n, _ in nat.named_parameters() if n.endswith("W_down"))
    assert sd[name].shape == dict(nat.named_parameters())[name].shape and len(sd) == 2 * len(tr.bucket.blocks)


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
def test_ema_matches_the_reference_class(backend):
    """The same three optimizer steps tracked by the reference's OWN ModelEMA (hcpdiff/utils/ema.py, loaded as a file: it only
    imports torch / numpy) and by hcp_ema_update: every averaged LoRA tensor agrees to fp32 rounding."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_hcp_ref_ema", "/root/reference/hcpdiff/utils/ema.py")
    ref_ema = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ema)
    _, nat = _pair(MICRO_CONFIG, backend.device)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-2, ema=dict(decay_max=0.95, inv_gamma=2.0, power=0.75))
    theirs = ref_ema.ModelEMA(nat, decay_max=0.95, inv_gamma=2.0, power=0.75)       # snapshots the requires_grad parameters (= the LoRA factors)
    theirs.train_params = {k: v.clone() for k, v in theirs.train_params.items()}    # its p.data.to('cpu') copies from a GPU model but ALIASES a CPU one
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        tr.bucket.grads.copy_(backend.to(torch.randn(tr.bucket.numel, generator=g)))
        tr.optimizer_step()
        theirs.update(nat)
    ours, ref = tr.ema_state_dict(), theirs.state_dict()
    lora_names = [n for n in ref if "lora_block_" in n and not n.endswith("alpha")]
    assert sorted(lora_names) == sorted(ours)
    for n in lora_names:
        assert torch.allclose(ours[n].cpu(), ref[n].cpu(), rtol=1e-5, atol=1e-7), n


@pytest.mark.parametrize("cfg_name", ["sd15", "sdxl"])
def test_from_pretrained_reads_diffusers_layout(tmp_path, cfg_name):
    """`model.unet: {_target_: ...NativeUNet2DConditionModel.from_pretrained, path, subfolder: unet}`:
    diffusers' on-disk layout (config.json with its historical key names + diffusion_pytorch_model.safetensors)."""
    from safetensors.torch import save_file
    cfg = TINY_CONFIG if cfg_name == "sd15" else TINY_SDXL_CONFIG
    ora = seeded_init_(OracleUNet2DConditionModel(**cfg), 3)
    d = tmp_path / "unet"
    d.mkdir()
    heads = cfg["num_attention_heads"]
    disk = dict(_class_name="UNet2DConditionModel", in_channels=4, out_channels=4, block_out_channels=list(cfg["block_out_channels"]),
                layers_per_block=cfg["layers_per_block"], down_block_types=list(cfg["down_block_types"]), up_block_types=list(cfg["up_block_types"]),
                attention_head_dim=heads if isinstance(heads, int) else list(heads),          # diffusers stores the head COUNT under this name
                cross_attention_dim=cfg["cross_attention_dim"], norm_num_groups=cfg["norm_num_groups"],
                use_linear_projection=cfg["use_linear_projection"], addition_embed_type=cfg["addition_embed_type"],
                addition_time_embed_dim=cfg["addition_time_embed_dim"],
                projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"])
    tl = cfg["transformer_layers_per_block"]
    disk["transformer_layers_per_block"] = tl if isinstance(tl, int) else list(tl)
    json.dump(disk, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in ora.state_dict().items()}, str(d / "diffusion_pytorch_model.safetensors"))
    nat = NativeUNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    sd = nat.state_dict()
    assert set(sd) == set(ora.state_dict()) and all(torch.equal(sd[k], v) for k, v in ora.state_dict().items())
    assert nat.config.cross_attention_dim == cfg["cross_attention_dim"] and nat.dtype == torch.float32


@pytest.mark.parametrize("rank", [48, 128, 200, 0.5])
def test_lora_rank_above_one_slot_group(backend, rank):
    """rank > 32 on a Linear host (any rank; 0.5 = the reference's fractional form, half the layer width: lora_base_patch.py:105-106):
    skinny side GEMM + K-extension instead of the fused 32-slot form; same reference arithmetic."""
    from hcp_diffusion_amd.layers import HipLinear
    dev = backend.device
    torch.manual_seed(int(rank * 10))
    nout = 40 if rank != 0.5 else 80
    parent = torch.nn.Module(); parent.fc = HipLinear(72, nout).to(dev)
    parent.requires_grad_(False)
    blk = LoraHipLayer.wrap_model(0, parent.fc, parent_block=parent, host_name="fc", rank=rank, alpha=8.0)[""]
    if rank == 0.5:
        rank = 40
    assert tuple(blk.layer.W_down.shape) == (rank, 72) and abs(float(blk.alpha) - 8.0 / rank) < 1e-7
    with torch.no_grad():
        blk.layer.W_up.normal_(0, 0.1)
    x = torch.randn(6, 9, 72).to(torch.bfloat16); dy = torch.randn(6, 9, nout).to(torch.bfloat16)
    wd, wu = (t.detach().cpu().clone().requires_grad_(True) for t in (blk.layer.W_down, blk.layer.W_up))
    xr = x.float().requires_grad_(True)
    yr = xr @ (parent.fc._host.weight.cpu() + float(blk.alpha) * (wu @ wd)).T + parent.fc._host.bias.cpu()
    yr.backward(dy.float())
    xn = backend.to(x).requires_grad_(True)
    y = parent.fc(xn)
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach(), yr.detach()) < 2e-2 and rel(xn.grad, xr.grad) < 2e-2
    assert rel(blk.layer.W_down.grad, wd.grad) < 2e-2 and rel(blk.layer.W_up.grad, wu.grad) < 2e-2


@pytest.mark.parametrize("rank,stride", [(40, 1), (72, 2)])
def test_conv_lora_rank_above_one_slot_group(backend, rank, stride):
    """rank > 32 on a 3x3 conv host (LoCon with a wide rank, lora_layers_patch.py:64-100): T = conv3x3(x, W_down) with ceil(rank/32)*32
    output channels, one more GEMM adds T (alpha W_up)^T to the host convolution; gradients through the same wgrad / dgrad kernels."""
    from hcp_diffusion_amd.layers import HipConv2d
    dev = backend.device
    torch.manual_seed(rank)
    cin, cout = 16, 24
    parent = torch.nn.Module(); parent.conv = HipConv2d(cin, cout, 3, stride, 1).to(dev)
    parent.requires_grad_(False)
    blk = LoraHipLayer.wrap_model(0, parent.conv, parent_block=parent, host_name="conv", rank=rank, alpha=8.0)[""]
    assert tuple(blk.layer.W_down.shape) == (rank, cin, 3, 3) and blk.wide
    with torch.no_grad():
        blk.layer.W_up.normal_(0, 0.1)
    x = torch.randn(2, 6, 6, cin).to(torch.bfloat16)
    Ho = 6 // stride
    dy = torch.randn(2, Ho, Ho, cout).to(torch.bfloat16)
    res = torch.randn(2, Ho, Ho, cout).to(torch.bfloat16)
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wd, wu = (t.detach().cpu().clone().requires_grad_(True) for t in (blk.layer.W_down, blk.layer.W_up))
    host = parent.conv._host
    w_eff = host.weight.detach().cpu().float() + float(blk.alpha) * torch.einsum("or,rikl->oikl", wu[:, :, 0, 0], wd)
    yr = torch.nn.functional.conv2d(xr, w_eff, host.bias.detach().cpu().float(), stride=stride, padding=1) + res.float().permute(0, 3, 1, 2)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    xn = backend.to(x).requires_grad_(True)
    y = parent.conv(xn, residual=backend.to(res))
    y.backward(backend.to(dy))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < 2e-2 and rel(xn.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
    assert rel(blk.layer.W_down.grad, wd.grad) < 2e-2 and rel(blk.layer.W_up.grad, wu.grad) < 2e-2


@pytest.mark.parametrize("which", ["conv_in", "conv_out"])
def test_lora_on_conv_in_and_conv_out(backend, which):
    """LoRA on the UNet's first / last convolution (4 latent channels: not a shape of the side-path kernels): the reference's merged-weight
    arithmetic (lora_base_patch.py:20-35) through the host's own kernels, TWO stacked blocks, against fp32 autograd on the merged weight."""
    from hcp_diffusion_amd.layers import HipConvIn, HipConvOut
    dev = backend.device
    torch.manual_seed(3)
    cin, cout = (4, 16) if which == "conv_in" else (16, 4)
    parent = torch.nn.Module(); parent.c = (HipConvIn if which == "conv_in" else HipConvOut)(cin, cout, 3, 1, 1).to(dev)
    parent.requires_grad_(False)
    b0 = LoraHipLayer.wrap_model(0, parent.c, parent_block=parent, host_name="c", rank=4, alpha=1.0)[""]
    b1 = LoraHipLayer.wrap_model(1, parent.c, parent_block=parent, host_name="c", rank=2, alpha=3.0)[""]
    assert b0.merged and parent.c.plugin_names == ["lora_block_0", "lora_block_1"] and tuple(b0.layer.W_down.shape) == (4, cin, 3, 3)
    with torch.no_grad():
        b0.layer.W_up.normal_(0, 0.1); b1.layer.W_up.normal_(0, 0.1)
    host = parent.c._host
    fac = [(b.layer.W_down.detach().cpu().clone().requires_grad_(True), b.layer.W_up.detach().cpu().clone().requires_grad_(True)) for b in (b0, b1)]
    w_eff = host.weight.detach().cpu().float() + sum(float(b.alpha) * torch.einsum("or,rikl->oikl", wu[:, :, 0, 0], wd) for b, (wd, wu) in zip((b0, b1), fac))
    x = torch.randn(2, cin, 8, 8)
    dy = torch.randn(2, cout, 8, 8)
    xr = x.clone().requires_grad_(which == "conv_out")
    yr = torch.nn.functional.conv2d(xr.to(torch.bfloat16).float() if which == "conv_in" else xr, w_eff, host.bias.detach().cpu().float(), padding=1)
    yr.backward(dy if which == "conv_out" else dy.to(torch.bfloat16).float())
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    if which == "conv_in":                       # NCHW latents in, NHWC bf16 activations out; the latents carry no gradient
        y = parent.c(backend.to(x))
        y.backward(backend.to(dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)))
        assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < 2e-2
    else:                                        # NHWC bf16 in, NCHW fp32 `.sample` out
        xn = backend.to(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)).requires_grad_(True)
        xr2 = xn.detach().float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr2, w_eff.detach(), host.bias.detach().cpu().float(), padding=1)
        for wd, wu in fac:
            wd.grad = None; wu.grad = None
        w2 = host.weight.detach().cpu().float() + sum(float(b.alpha) * torch.einsum("or,rikl->oikl", wu[:, :, 0, 0], wd) for b, (wd, wu) in zip((b0, b1), fac))
        yr = torch.nn.functional.conv2d(xr2, w2, host.bias.detach().cpu().float(), padding=1)
        yr.backward(dy)
        y = parent.c(xn)
        y.backward(backend.to(dy))
        assert rel(y.detach(), yr.detach()) < 2e-2 and rel(xn.grad.permute(0, 3, 1, 2), xr2.grad) < 2e-2
    for b, (wd, wu) in zip((b0, b1), fac):
        assert rel(b.layer.W_down.grad, wd.grad) < 3e-2 and rel(b.layer.W_up.grad, wu.grad) < 3e-2
    assert host.weight.grad is None and not host.weight.requires_grad        # the frozen host stays untouched


@pytest.mark.parametrize("host_mode", ["trainable", "bf16"])
def test_merged_lora_host_gradient_and_bf16_host(backend, host_mode):
    """the merged-weight fallback (a) hands dW_eff to a TRAINABLE host weight as well (the reference differentiates
    layer(x, host_weight + weight), lora_base_patch.py:20-35: full fine-tune + LoRA on conv_in trains both), (b) builds its shadow weight
    in fp32 so that a frozen bf16 host (the reference casts TE_unet to weight_dtype) does not trip the fp32 gradient buffer, and (c) a
    gradient-free call (the sampler) reuses the merged weight until a factor changes."""
    from hcp_diffusion_amd.layers import HipConvIn
    dev = backend.device
    torch.manual_seed(5)
    cin, cout = 4, 16
    parent = torch.nn.Module(); parent.c = HipConvIn(cin, cout, 3, 1, 1).to(dev)
    parent.requires_grad_(host_mode == "trainable")
    if host_mode == "bf16":
        parent.c.to(torch.bfloat16)
    blk = LoraHipLayer.wrap_model(0, parent.c, parent_block=parent, host_name="c", rank=4, alpha=2.0)[""]
    with torch.no_grad():
        blk.layer.W_up.normal_(0, 0.1)
    host = parent.c._host
    wd = blk.layer.W_down.detach().cpu().float().clone().requires_grad_(True)
    wu = blk.layer.W_up.detach().cpu().float().clone().requires_grad_(True)
    hw = host.weight.detach().cpu().float().clone().requires_grad_(True)
    x = torch.randn(2, cin, 8, 8)
    dy = torch.randn(2, cout, 8, 8).to(torch.bfloat16).float()
    yr = torch.nn.functional.conv2d(x.to(torch.bfloat16).float(), hw + float(blk.alpha) * torch.einsum("or,rikl->oikl", wu[:, :, 0, 0], wd),
                                    host.bias.detach().cpu().float(), padding=1)
    yr.backward(dy)
    y = parent.c(backend.to(x))
    y.backward(backend.to(dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)))
    rel = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < 2e-2
    assert rel(blk.layer.W_down.grad, wd.grad) < 3e-2 and rel(blk.layer.W_up.grad, wu.grad) < 3e-2
    if host_mode == "trainable":
        assert host.weight.grad is not None and rel(host.weight.grad, hw.grad) < 3e-2
        assert host.bias.grad is not None and rel(host.bias.grad, dy.sum((0, 2, 3))) < 3e-2
    else:
        assert host.weight.grad is None
    with torch.no_grad():                       # gradient-free calls: one merge, then cache hits until a factor changes
        y1 = parent.c(backend.to(x)); y2 = parent.c(backend.to(x))
        assert torch.equal(y1, y2) and rel(y1.permute(0, 3, 1, 2), yr.detach()) < 2e-2
        blk.layer.W_up.mul_(2.0)
        y3 = parent.c(backend.to(x))
        assert not torch.equal(y1, y3)
        # the trainer's optimizer writes the factors through raw pointers (hcp_adamw_clip_fused) — no torch version counter
        # moves; LoraBucket.pack() (and NativeTrainer after every step) invalidate the cached merge
        from hcp_diffusion_amd.lora import LoraBucket
        bucket = LoraBucket([blk])
        y4 = parent.c(backend.to(x))
        n = bucket.params.numel()
        g = torch.ones(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
        lr = torch.tensor([0.1], device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev)
        before = bucket.params.clone()
        ver = bucket.params._version
        K.adamw_clip_fused(bucket.params, g, m, v, lr, step)
        assert bucket.params._version == ver and not torch.equal(before, bucket.params), "the raw-pointer step moved the parameters unseen"
        bucket.pack()
        y5 = parent.c(backend.to(x))
        assert not torch.equal(y4, y5), "a gradient-free call after an optimizer step must see the new factors"


def test_lora_dropout_and_svd_init(backend):
    """dropout > 0: the reference drops the whole layer output (lora_base_patch.py:74) — eval mode is the identity, train mode
    zeroes ~p of the outputs and rescales the rest by 1/(1-p), gradients flow through the kept ones only.  svd_init: the factors
    start as the clamped rank-r SVD of the host weight (lora_base_patch.py:76-82, utils/utils.py:17-41 — the Linear branch of the
    reference's own low_rank_approximate is the oracle where the reference tree exists)."""
    from hcp_diffusion_amd.layers import HipLinear
    dev = backend.device
    torch.manual_seed(3)
    lin = HipLinear(64, 96, bias=True).to(dev)
    parent = torch.nn.Module(); parent.proj = lin
    blk = LoraHipLayer.wrap_layer(0, lin, rank=4, dropout=0.5, parent_block=parent, host_name="proj")
    with torch.no_grad():
        blk.layer.W_up.normal_(0, 0.05)
    x = backend.to(torch.randn(2, 40, 64).to(torch.bfloat16))
    parent.proj.eval(); blk.eval()
    y_eval = parent.proj(x).float()
    blk.dropout.p = 0.0
    y_ref = parent.proj(x).float()
    assert torch.equal(y_eval, y_ref)
    blk.dropout.p = 0.5
    blk.train()
    xg = x.clone().requires_grad_(True)
    y = parent.proj(xg)
    yf = y.float()
    dropped = (yf == 0).float().mean().item()
    assert 0.4 < dropped < 0.6
    kept = yf != 0
    assert ((yf[kept] - 2.0 * y_ref[kept]).abs().max() / y_ref.abs().max()).item() < 2e-2
    y.float().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad.float()).all() and blk.layer.W_up.grad.abs().sum().item() > 0
    # svd_init
    lin2 = HipLinear(64, 96, bias=False).to(dev)
    parent2 = torch.nn.Module(); parent2.proj = lin2
    blk2 = LoraHipLayer.wrap_layer(1, lin2, rank=8, svd_init=True, parent_block=parent2, host_name="proj")
    w = lin2.weight.detach().float().cpu()
    U, S, Vh = torch.linalg.svd(w, full_matrices=False)
    best = (U[:, :8] * S[:8]) @ Vh[:8]
    approx = (blk2.layer.W_up.detach().float().cpu().reshape(96, 8) @ blk2.layer.W_down.detach().float().cpu().reshape(8, 64))
    assert ((approx - best).norm() / best.norm()).item() < 0.15         # the 0.99-quantile clamp trims the largest entries only
    if os.path.isdir("/root/reference/hcpdiff"):
        from oracle.ref_shims import load_reference_lora
        load_reference_lora()
        from hcpdiff.utils.utils import low_rank_approximate
        Ur, Vr = low_rank_approximate(w, 8)
        # singular vectors are defined up to sign per component: compare the products
        assert ((approx - Ur @ Vr).norm() / (Ur @ Vr).norm()).item() < 1e-4


def test_controlnet_branch_from_a_lora_wrapped_host(backend):
    """copy_block (controlnet.py:38-44): the branch copies the PLAIN host layers even when the host already carries LoRA blocks."""
    from hcp_diffusion_amd.controlnet import make_controlnet
    from hcp_diffusion_amd.lora import LoraHipContainer
    dev = backend.device
    _, nat = _pair(MICRO_CONFIG, dev)
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-3)
    assert any(isinstance(m, LoraHipContainer) for m in nat.modules())
    plug = make_controlnet(nat, block_out_channels=MICRO_CONFIG["block_out_channels"], layers_per_block=MICRO_CONFIG["layers_per_block"],
                           cond_block_channels=(3, 8, 8, 16, 16, MICRO_CONFIG["block_out_channels"][0]))
    assert not any(isinstance(m, LoraHipContainer) for m in plug.modules())
    assert not any("lora_block" in n for n, _ in plug.named_parameters())
    n_host = sum(p.numel() for n, p in nat.down_blocks.named_parameters() if "lora_block" not in n)
    assert sum(p.numel() for p in plug.down_blocks.parameters()) == n_host


@pytest.mark.parametrize("variant", ["lora", "lora_masked", "frozen_hosts"])
def test_batched_cross_attention_kv_matches_the_per_layer_projections(backend, variant):
    """Every cross-attention layer reads the same prompt states: unet._batched_ctx_kv projects all their keys / values in three launches
    (lora.CtxBatch: T_all = ctx AD_all^T, [ctx | T_all] [W_all | BU]^T) and the attention backward writes the K/V gradients into slices of
    ONE buffer.  Same prediction and the same LoRA gradients as one fused launch pair per layer — with LoRA on the projections, with an
    encoder_attention_mask (key-bias attention path), and with bare frozen hosts (LoRA elsewhere only)."""
    from hcp_diffusion_amd import kernels as Kn
    dev = backend.device
    pats = PATS if variant != "frozen_hosts" else [r"re:.*\.ff$"]
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 4, 8, 8, generator=g); ehs = torch.randn(2, 24, 32, generator=g)
    noise = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([20, 700])
    mask = None
    if variant == "lora_masked":
        mask = torch.ones(2, 24); mask[0, 10:] = 0
    res = {}
    for batched in (True, False):
        _, nat = _pair(MICRO_CONFIG, dev)                        # 4 cross-attention layers
        tr = NativeTrainer(nat, [dict(layers=pats, rank=4)], lr=1e-3)
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for blk in tr.bucket.blocks:
                blk.layer.W_up.copy_((torch.randn(blk.layer.W_up.shape, generator=gen) * 0.05).to(dev))
        tr.bucket.pack()
        if not batched:
            nat._batched_ctx_kv = lambda ctx: None
        tr.make_noise = lambda lat: (Kn.add_noise(lat, backend.to(noise), backend.to(t), tr.acp), backend.to(noise), backend.to(t))
        Kn.TRACE = []
        try:
            loss = tr.forward_backward(backend.to(x0), backend.to(ehs), attn_mask=backend.to(mask) if mask is not None else None)
            trace = list(Kn.TRACE)
        finally:
            Kn.TRACE = None
        res[batched] = (loss.item(), tr.bucket.grads.detach().float().cpu().clone(), len(trace))
        if batched:
            n_layers = sum(1 for n, _ in nat.named_modules() if n.endswith(".attn2"))
            assert nat._ctx_batch[1].n_total > 0 and len(nat._ctx_batch[1].groups) == n_layers
    (lb, gb, nb), (lp, gp, np_) = res[True], res[False]
    assert abs(lb - lp) <= 2e-3 * abs(lp)
    assert torch.nn.functional.cosine_similarity(gb, gp, dim=0).item() > 0.9995 and abs(gb.norm() - gp.norm()) / gp.norm() < 1e-2
    assert nb < np_                                                   # fewer GEMM launches
