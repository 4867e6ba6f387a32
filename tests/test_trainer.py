"""NativeTrainer behaviours added in round 2: per-item learning rates (cfg_net_tools.py:108-123), gradient accumulation
(train_ac.py:119,468), loss.type 'sample' (train_ac.py:458-465), and — on the GPU — hipGraph mode: the capture must not
perturb the training state, two datasets per step must replay their own descriptor tables, and a new latent shape
(aspect-ratio bucket, data/bucket.py:167-204) gets its own captured graph."""
import pytest
import torch

from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_

PATS_A, PATS_F = [r"re:.*\.attn.?$"], [r"re:.*\.ff$"]


def _native(dev):
    torch.manual_seed(0)
    ora = seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1)
    nat = NativeUNet2DConditionModel(**MICRO_CONFIG)
    nat.load_state_dict(ora.state_dict())
    return nat.to(dev)


def _trainer(dev, lora_cfg=None, **kw):
    tr = NativeTrainer(_native(dev), lora_cfg or [dict(layers=PATS_A + PATS_F, rank=4)], lr=kw.pop("lr", 1e-2), **kw)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in tr.bucket.blocks:
            blk.layer.W_up.copy_((torch.randn(blk.layer.W_up.shape, generator=g) * 0.05).to(dev))
    tr.bucket.pack()
    return tr


def _batch(dev, seed, hw=(8, 8), B=1, L=24):
    g = torch.Generator().manual_seed(seed)
    return dict(latents=torch.randn(B, 4, *hw, generator=g).to(dev), encoder_hidden_states=torch.randn(B, L, 32, generator=g).to(dev))


def _fix_noise(tr, dev, seed=11):
    """Deterministic make_noise (same draw in every call for a given latent shape)."""
    cache = {}                                                        # device tensors made once per shape: capture-safe afterwards

    def mk(lat):
        key = tuple(lat.shape)
        if key not in cache:
            g = torch.Generator().manual_seed(seed + lat.shape[2] * 131 + lat.shape[3])
            cache[key] = (torch.randn(lat.shape, generator=g).to(dev), torch.randint(0, 1000, (lat.shape[0],), generator=g).to(dev))
        n, t = cache[key]
        return K.add_noise(lat, n, t, tr.acp), n, t
    tr.make_noise = mk


def test_per_item_learning_rates(backend):
    """Two lora_unet items with different lr: each item's parameters move with ITS lr (one AdamW param group per item)."""
    dev = backend.device
    tr = _trainer(dev, [dict(layers=PATS_A, rank=4, lr=1e-2), dict(layers=PATS_F, rank=4, lr=1e-4)], lr=1e-3)
    st = tr._lora_state
    assert len(st.segments) == 2 and st.base_lrs == [1e-2, 1e-4]
    _fix_noise(tr, dev)
    p0 = tr.bucket.params.clone()
    b = _batch(dev, 1)
    tr.train_one_step(b["latents"], b["encoder_hidden_states"])
    d = (tr.bucket.params - p0).abs()
    (o0, n0), (o1, n1) = st.segments
    m0, m1 = d[o0:o0 + n0].max().item(), d[o1:o1 + n1].max().item()
    assert 0.5e-2 < m0 < 1.3e-2 and 0.5e-4 < m1 < 1.3e-4, (m0, m1)      # first AdamW step moves a parameter by ~lr
    tr.set_lr_factor(0.5)                                             # scheduler: every group keeps its own base lr
    assert abs(st.lrs[0].item() - 0.5e-2) < 1e-9 and abs(st.lrs[1].item() - 0.5e-4) < 1e-9


def test_gradient_accumulation_matches_one_big_step(backend):
    """accelerator.accumulate (train_ac.py:119,468): two micro-steps of one sample each == one step on both samples (mean loss)."""
    dev = backend.device
    b1, b2 = _batch(dev, 1), _batch(dev, 2)
    n = [torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(7 + i)).to(dev) for i in range(2)]
    t = [torch.tensor([100]).to(dev), torch.tensor([700]).to(dev)]
    tr = _trainer(dev, gradient_accumulation_steps=2)
    seq = iter([0, 1])
    tr.make_noise = lambda lat: (lambda i: (K.add_noise(lat, n[i], t[i], tr.acp), n[i], t[i]))(next(seq))
    p0 = tr.bucket.params.clone()
    tr.train_one_step(b1["latents"], b1["encoder_hidden_states"])
    assert torch.equal(tr.bucket.params, p0) and tr.bucket.grads.abs().max().item() > 0     # no optimizer step yet
    tr.train_one_step(b2["latents"], b2["encoder_hidden_states"])
    assert tr.bucket.grads.abs().max().item() == 0.0 and not torch.equal(tr.bucket.params, p0)
    ref = _trainer(dev)
    nn_, tt = torch.cat(n), torch.cat(t)
    ref.make_noise = lambda lat: (K.add_noise(lat, nn_, tt, ref.acp), nn_, tt)
    ref.train_one_step(torch.cat([b1["latents"], b2["latents"]]), torch.cat([b1["encoder_hidden_states"], b2["encoder_hidden_states"]]))
    assert ((tr.bucket.params - ref.bucket.params).abs().max() / ref.bucket.params.abs().max()).item() < 2e-3


def test_loss_type_sample_is_the_reweighted_eps_loss(backend):
    """loss.type 'sample' (train_ac.py:460-463): MSE between the x0 recovered from the prediction and from the true noise
    = (1 - acp_t) / acp_t * MSE(eps_hat, eps) per sample."""
    dev = backend.device
    b = _batch(dev, 3, B=2)
    tr_e, tr_s = _trainer(dev), _trainer(dev, loss_type="sample")
    for tr in (tr_e, tr_s):
        _fix_noise(tr, dev)
    tr_e.forward_backward(b["latents"][:1], b["encoder_hidden_states"][:1])
    ls = tr_s.forward_backward(b["latents"][:1], b["encoder_hidden_states"][:1])
    _, _, t = tr_e.make_noise(b["latents"][:1])
    a = tr_e.acp[t].item()
    w = (1 - a) / a
    ge, gs = w * tr_e.bucket.grads, tr_s.bucket.grads                # (same bf16 backward on a rescaled dL/dpred: equal to rounding)
    assert torch.nn.functional.cosine_similarity(ge, gs, dim=0).item() > 0.9999 and abs(gs.norm().item() / ge.norm().item() - 1) < 1e-2
    assert ls.item() > 0
    with pytest.raises(ValueError):
        NativeTrainer(_native(dev), [dict(layers=PATS_A, rank=4)], loss_type="v")


def test_train_one_step_takes_attn_mask_and_leaves_the_batch_alone(backend):
    dev = backend.device
    tr = _trainer(dev)
    _fix_noise(tr, dev)
    b = _batch(dev, 4, B=2)
    lat16 = b["latents"].half()
    batch = dict(latents=lat16, encoder_hidden_states=b["encoder_hidden_states"])
    tr.train_data_list([batch])
    assert batch["latents"] is lat16                                  # the caller's dict is not rewritten
    mask = torch.ones(2, 24); mask[:, 20:] = 0
    l1 = tr.train_one_step(b["latents"], b["encoder_hidden_states"], attn_mask=backend.to(mask))
    assert torch.isfinite(l1).all()


# ------------------------------------------------------------------------------------------------ hipGraph mode (GPU)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    K._set_backend_for_tests(None)
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_graph_capture_does_not_perturb_training_state():
    """The first step in hipGraph mode is the FIRST optimisation step: parameters, moments, step counters and the RNG stream
    after it equal eager mode's (the capture's warm-up steps are rolled back)."""
    dev = _gpu()
    b = _batch(dev, 1, B=2)
    outs = []
    for use_graph in (False, True):
        tr = _trainer(dev, use_graph=use_graph, ema=dict(decay_max=0.99))
        torch.manual_seed(123); torch.cuda.manual_seed(123)           # make_noise draws from torch's generators
        for _ in range(2):
            tr.train_one_step(b["latents"], b["encoder_hidden_states"])
        torch.cuda.synchronize()
        outs.append((tr.bucket.params.clone(), tr.exp_avg.clone(), tr.step_count.item(), tr._lora_state.ema.clone(), torch.rand(1, device=dev).item()))
    (pe, me, se, ee, re_), (pg, mg, sg, eg, rg) = outs
    assert se == sg == 2
    assert ((pe - pg).abs().max() / pe.abs().max()).item() < 1e-5 and ((me - mg).abs().max() / me.abs().max()).item() < 1e-4
    assert ((ee - eg).abs().max() / ee.abs().max()).item() < 1e-5
    assert re_ == rg                                                  # same position in the device RNG stream


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", [True, False])
def test_graph_two_datasets_equal_eager(grouped):
    """DreamBooth instance + class batch in one captured step: both forward/backward passes replay their OWN LoRA weight-gradient
    descriptor tables (same byte length: the staging buffers must not be shared), with one grouped launch or one per layer."""
    dev = _gpu()
    data = [dict(**_batch(dev, 1)), dict(**_batch(dev, 2), loss_weight=0.5)]
    res = []
    for use_graph in (False, True):
        tr = _trainer(dev, use_graph=use_graph, grouped_wgrad=grouped)
        _fix_noise(tr, dev)
        tr.optimizer_step_real, grads = tr.optimizer_step, []
        for step in range(3):
            tr.optimizer_step = lambda: None
            tr._opt_graph = None
            tr.train_data_list([dict(d) for d in data])
            torch.cuda.synchronize()
            grads.append(tr.bucket.grads.clone())
            tr.bucket.grads.zero_()
        res.append(grads)
    for ge, gg in zip(*res):
        assert ((ge - gg).norm() / ge.norm()).item() < 1e-5


@pytest.mark.gpu
def test_graph_cache_per_latent_shape():
    """Aspect-ratio buckets: steps alternate between two resolutions (and context lengths); each gets its own captured graph and
    the trajectory equals eager mode's."""
    dev = _gpu()
    shapes = [((8, 8), 24), ((8, 16), 24), ((16, 8), 48)]
    batches = [_batch(dev, 10 + i, hw=hw, B=2, L=L) for i, (hw, L) in enumerate(shapes)]
    order = [0, 1, 0, 2, 1, 0]
    outs = []
    for use_graph in (False, True):
        tr = _trainer(dev, use_graph=use_graph, lr=1e-3)
        _fix_noise(tr, dev)
        for i in order:
            tr.train_one_step(batches[i]["latents"], batches[i]["encoder_hidden_states"])
        torch.cuda.synchronize()
        outs.append(tr.bucket.params.clone())
        if use_graph:
            assert len(tr._graph_cache) == 3
    assert ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item() < 1e-4


@pytest.mark.gpu
def test_graph_trainer_300_step_soak_tracks_eager_on_the_real_architecture():
    """The test that would have caught the round-3 NaN (a memset node in front of an atomics kernel, wrong from SOME replay on): the
    captured step of NativeTrainer replayed 300 times on the full SD1.5 architecture at 64x64 latents — the only size at which the
    query-split cross-attention backward, the split-K GEMMs and the multi-range LoRA weight gradients all run — against the same 300
    steps eager, same weights, same noise / timestep stream.  Every 50th loss within 1e-3 (relative), final LoRA within 1e-3 of the
    update's norm... measured on MI355X: see the message printed with -rP."""
    dev = _gpu()
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel as U
    B, steps, every = 2, 300, 50
    res = []
    for use_graph in (False, True):
        torch.manual_seed(7)
        with torch.device("meta"):
            unet = U()
        unet = unet.to_empty(device=dev)
        g = torch.Generator(device=dev).manual_seed(99)
        with torch.no_grad():
            for name, p in unet.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g, device=dev) * p[0].numel() ** -0.5)
                elif "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
        tr = NativeTrainer(unet, [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=8, lr=1e-4)], lr=1e-4, use_graph=use_graph)
        with torch.no_grad():
            for blk in tr.bucket.blocks:
                blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g, device=dev) * 0.02)
        tr.bucket.pack()
        p0 = tr.bucket.params.clone()
        lat = torch.randn(B, 4, 64, 64, generator=g, device=dev)
        ehs = torch.randn(B, 77, 768, generator=g, device=dev).to(torch.bfloat16)
        torch.manual_seed(123); torch.cuda.manual_seed(123)           # make_noise draws from torch's device generator
        losses = []
        for i in range(steps):
            loss = tr.train_one_step(lat, ehs)
            if (i + 1) % every == 0:
                losses.append(float(loss.item()))
        torch.cuda.synchronize()
        res.append((losses, (tr.bucket.params - p0).clone()))
        del tr, unet
        torch.cuda.empty_cache()
    (le, de), (lg, dg) = res
    rel = [abs(a - b) / abs(a) for a, b in zip(le, lg)]
    upd = ((de - dg).norm() / de.norm()).item()
    print(f"soak: eager losses {[round(v, 5) for v in le]}  graph losses {[round(v, 5) for v in lg]}  max rel {max(rel):.2e}  LoRA update diff {upd:.2e}")
    assert all(v == v and abs(v) != float("inf") for v in le + lg)
    assert max(rel) < 1e-3, (le, lg)
    assert upd < 2e-2, upd


@pytest.mark.gpu
def test_unet_hip_graph_under_a_plain_trainer_loop_matches_eager():
    """unet.enable_hip_graph(): an ordinary eager loop (module call, loss.backward(), clip, torch AdamW, zero_grad) — what the
    reference's Trainer.train_one_step does — replays captured forward / backward graphs and follows the eager trajectory; the
    capture leaves gradients untouched; zero_grad(set_to_none=True) is survived; a second input shape gets its own graphs."""
    from hcp_diffusion_amd.lora import make_lora
    dev = torch.device("cuda:0")

    def run(graph, steps=5):
        unet = _native(dev).requires_grad_(False)                   # the host is frozen, as the reference trainer leaves it (train_ac.py:232-240)
        groups, grp, bucket = make_lora(unet, [dict(layers=PATS_A + PATS_F, rank=4, lr=1e-2)])
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for blk in bucket.blocks:
                blk.layer.W_up.copy_((torch.randn(blk.layer.W_up.shape, generator=g) * 0.05).to(dev))
        bucket.pack()
        params = [p for blk in bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
        opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=1e-3)
        if graph:
            unet.enable_hip_graph()
        losses = []
        for i in range(steps):
            hw = (8, 8) if i != 3 else (8, 16)                       # step 3: another aspect-ratio bucket
            b = _batch(dev, 100 + i, hw=hw, B=2)
            t = torch.tensor([10 + 100 * i, 900 - 50 * i], device=dev)
            target = torch.randn(2, 4, *hw, generator=torch.Generator().manual_seed(i)).to(dev)
            pred = unet(b["latents"], t, b["encoder_hidden_states"].to(torch.bfloat16)).sample
            loss = torch.nn.functional.mse_loss(pred.float(), target)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            opt.zero_grad(set_to_none=(i % 2 == 1))
            losses.append(loss.item())
        if graph:
            assert len(unet._hip_graphs) == 2
        return losses, torch.cat([p.detach().flatten() for p in params]).cpu()

    le, pe = run(False)
    lg, pg = run(True)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (le, lg)
    assert ((pe - pg).abs().max() / pe.abs().max()).item() < 5e-3
    assert le[0] != le[-1]
    # two forwards before one backward (losses summed): the second call must not overwrite the first one's saved activations
    unet = _native(dev).requires_grad_(False)
    _, _, bucket = make_lora(unet, [dict(layers=PATS_A, rank=4)])
    with torch.no_grad():
        for blk in bucket.blocks:
            blk.layer.W_up.normal_(0, 0.05)
    bucket.pack()

    def two(graph):
        unet.enable_hip_graph(graph)
        bucket.grads.zero_()
        b1, b2 = _batch(dev, 1, B=2), _batch(dev, 2, B=2)
        t = torch.tensor([100, 800], device=dev)
        p1 = unet(b1["latents"], t, b1["encoder_hidden_states"].to(torch.bfloat16)).sample
        p2 = unet(b2["latents"], t, b2["encoder_hidden_states"].to(torch.bfloat16)).sample
        (p1.float().square().mean() + p2.float().square().mean()).backward()
        return bucket.grads.clone()
    ge, gg = two(False), two(True)
    assert torch.nn.functional.cosine_similarity(ge, gg, dim=0).item() > 0.9999


@pytest.mark.gpu
def test_text_encoder_and_unet_hip_graphs_under_a_plain_trainer_loop():
    """lora_conventional.yaml trains LoRA on the UNet AND the text encoder: with enable_hip_graph() on both modules the encoder's graph
    feeds the UNet's, the UNet's backward graph returns d(encoder_hidden_states) to the encoder's backward graph; same trajectory as eager."""
    from hcp_diffusion_amd.lora import make_lora
    from hcp_diffusion_amd.text_encoder import NativeCLIPTextModel
    from oracle.clip_ref import OracleCLIPTextModel
    dev = torch.device("cuda:0")
    tcfg = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    ucfg = dict(MICRO_CONFIG, cross_attention_dim=64)

    def run(graph, steps=4):
        torch.manual_seed(0)
        unet = NativeUNet2DConditionModel(**ucfg)
        unet.load_state_dict(seeded_init_(OracleUNet2DConditionModel(**ucfg), 1).state_dict())
        unet = unet.to(dev).requires_grad_(False)
        te = NativeCLIPTextModel(**tcfg)
        te.load_state_dict(seeded_init_(OracleCLIPTextModel(**tcfg), 2).state_dict())
        te = te.to(dev).requires_grad_(False)
        _, _, bu = make_lora(unet, [dict(layers=PATS_A + PATS_F, rank=4)])
        _, _, bt = make_lora(te, [dict(layers=[r"re:.*self_attn$", r"re:.*mlp$"], rank=4)])
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for bk in (bu, bt):
                for blk in bk.blocks:
                    blk.layer.W_up.copy_((torch.randn(blk.layer.W_up.shape, generator=g) * 0.05).to(dev))
                bk.pack()
        params = [p for bk in (bu, bt) for blk in bk.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
        opt = torch.optim.AdamW(params, lr=5e-3, weight_decay=1e-3)
        if graph:
            unet.enable_hip_graph(); te.enable_hip_graph()
        losses = []
        for i in range(steps):
            gi = torch.Generator().manual_seed(50 + i)
            ids = torch.randint(0, 100, (2, 24), generator=gi).to(dev)
            x = torch.randn(2, 4, 8, 8, generator=gi).to(dev); target = torch.randn(2, 4, 8, 8, generator=gi).to(dev)
            t = torch.tensor([30 + 200 * i, 950 - 100 * i], device=dev)
            ehs = te(ids, output_hidden_states=True)[0]
            loss = torch.nn.functional.mse_loss(unet(x, t, ehs).sample.float(), target)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step(); opt.zero_grad(set_to_none=False)
            losses.append(loss.item())
        if graph:
            assert len(unet._hip_graphs) == 1 and len(te._hip_graphs) == 1
        return losses, torch.cat([p.detach().flatten() for p in params]).cpu(), bt.params.detach().cpu().clone()

    le, pe, te_e = run(False)
    lg, pg, te_g = run(True)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (le, lg)
    assert ((pe - pg).abs().max() / pe.abs().max()).item() < 5e-3
    init = run(False, steps=0)[2]
    assert (te_g - init).abs().max().item() > 1e-3          # the text-encoder factors really trained through both graphs


def test_fused_adamw_state_dict_round_trip(backend):
    """FusedAdamW keeps flat moments per contiguous run; state_dict() cuts them into torch's per-parameter layout and load_state_dict()
    puts them back (resume must not silently reset the Adam moments): a resumed FusedAdamW continues exactly like the one that was
    saved, and a torch.optim.AdamW resumes from the same dict (train_ac.py:370 builds either from the same cfg)."""
    from hcp_diffusion_amd.optim import FusedAdamW
    dev = backend.device
    torch.manual_seed(0)
    flat = torch.randn(64 + 48, device=dev)
    ps = [torch.nn.Parameter(flat[:64].view(8, 8)), torch.nn.Parameter(flat[64:].view(6, 8))]
    gflat = torch.zeros_like(flat)
    for p, g in zip(ps, (gflat[:64].view(8, 8), gflat[64:].view(6, 8))):
        p.grad = g

    def grad(i):
        return torch.sin(torch.arange(gflat.numel(), device=dev) * (0.1 + i))

    a = FusedAdamW(ps, lr=1e-2)
    for i in range(3):
        gflat.copy_(grad(i)); a.step()
    sd = a.state_dict()
    assert set(sd["state"]) == {0, 1} and sd["state"][0]["exp_avg"].shape == (8, 8) and float(sd["state"][1]["step"]) == 3.0
    snapshot = flat.detach().clone()
    gflat.copy_(grad(3)); a.step()                     # the step the saved optimizer takes next
    expect = flat.detach().clone()
    # a fresh FusedAdamW resumed from the dict takes the same step from the same parameters
    with torch.no_grad():
        flat.copy_(snapshot)
    b = FusedAdamW(ps, lr=1e-2)
    b.load_state_dict(sd)
    gflat.copy_(grad(3)); b.step()
    assert torch.allclose(flat, expect, atol=1e-7), (flat - expect).abs().max()
    # ... and so does torch's own AdamW over copies of the parameters
    d_params = [torch.nn.Parameter(snapshot[:64].view(8, 8).clone()), torch.nn.Parameter(snapshot[64:].view(6, 8).clone())]
    d = torch.optim.AdamW(d_params, lr=1e-2)
    d.load_state_dict(sd)
    g3 = grad(3)
    d_params[0].grad, d_params[1].grad = g3[:64].view(8, 8).clone(), g3[64:].view(6, 8).clone()
    d.step()
    got = torch.cat([p.detach().flatten() for p in d_params])
    assert torch.allclose(got, expect, atol=1e-6), (got - expect).abs().max()


@pytest.mark.parametrize("use_graph", [False, True])
def test_sharded_exchange_from_backward_on_one_rank(backend, use_graph):
    """The sharded-optimizer machinery with everything on — chunks reduce-scattered from backward hooks on the side stream, bf16 wire
    casts, per-chunk AdamW launches — forced onto ONE rank (the collectives degenerate to copies), DreamBooth's two datasets per step:
    the same training as the plain full fine-tune, up to the bf16 rounding of the gradients on the wire.  On the GPU also inside the
    captured step: the side-stream branch must fork from and join the capture."""
    if use_graph and not backend.is_gpu:
        pytest.skip("hipGraph capture needs the GPU")
    dev = backend.device
    data = [dict(**_batch(dev, 1)), dict(**_batch(dev, 2), loss_weight=0.5)]
    res = {}
    for name, kw in (("plain", {}), ("fp32", dict(shard_optimizer="force", overlap_exchange=True)),
                     ("bf16", dict(shard_optimizer="force", overlap_exchange=True, grad_wire="bf16", param_wire="bf16"))):
        tr = NativeTrainer(_native(dev), None, lr=1e-3, train_cfg=[dict(layers=[""])], use_graph=use_graph, **kw)
        _fix_noise(tr, dev)
        for _ in range(2):
            tr.train_data_list([dict(d) for d in data])
        if kw:
            st = tr.host_buckets[0]
            assert st.shard and [p[0] for p in st.parts] == ([0, 1, 2, 12] if name == "bf16" else [0, 1, 2])
            assert tuple(sorted(tr._sent if not use_graph else next(iter(tr._graph_cache.values()))[3])) == (0, 1)
            assert all(s_.item() == 2 for s_ in st.steps)
            assert st.bucket.grads.abs().max().item() == 0
        res[name] = torch.cat([p.detach().float().flatten().cpu() for _, p in sorted(tr.unet.named_parameters())])
    init = torch.cat([p.detach().float().flatten() for _, p in sorted(_native("cpu").named_parameters())])
    moved = (res["plain"] - init).norm().item()
    assert moved > 0
    # Interpreter: deterministic, the fp32 variant is the same arithmetic (the clip norm adds the slices in another order: 1e-6).
    # GPU: the weight-gradient kernels accumulate with atomics, so two runs of the SAME trainer differ by ~1e-7 in the gradients, and
    # AdamW's normalised update turns that into O(lr) on elements whose gradient is ~0 — tools/diag/shard_race_diag.py: two plain runs
    # land in one of two outcomes 7e-3 of the distance moved apart (62 tensors, <= 5e-4 per element), whichever variant runs.
    assert (res["fp32"] - res["plain"]).norm().item() / moved < (2e-2 if backend.is_gpu else 1e-4)
    assert (res["bf16"] - res["plain"]).norm().item() / moved < 5e-2


def test_fused_adamw_zero_grad_clears_flat_runs_and_stragglers(backend):
    """zero_grad(set_to_none=False) of the seam optimizer: one fill per flat run (the bucket views) plus torch's way for a parameter that
    lives elsewhere; set_to_none=True stays torch's (views dropped)."""
    from hcp_diffusion_amd.optim import FusedAdamW
    dev = backend.device
    flat_p, flat_g = torch.randn(96, device=dev), torch.zeros(96, device=dev)
    ps = []
    for i in range(3):
        p = torch.nn.Parameter(torch.empty(0, device=dev))
        p.data = flat_p[32 * i:32 * i + 32].view(4, 8); p.grad = flat_g[32 * i:32 * i + 32].view(4, 8)
        ps.append(p)
    lone = torch.nn.Parameter(torch.randn(5, device=dev)); lone.grad = torch.zeros(5, device=dev)
    opt = FusedAdamW([dict(params=ps + [lone], lr=1e-2)])
    flat_g.fill_(1.0); lone.grad.fill_(1.0)
    opt.step()
    assert len(opt._runs[0][1]) == 2                       # the three views are ONE run, the lone parameter another
    flat_g.fill_(2.0); lone.grad.fill_(2.0)                 # a backward that ran after the step
    opt.zero_grad(set_to_none=False)
    assert flat_g.abs().max().item() == 0 and lone.grad.abs().max().item() == 0 and all(p.grad is not None for p in ps)
    opt.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in ps + [lone])


def test_bf16_param_wire_routes_only_hip_layer_weights_and_the_stale_flag_follows_the_sync_step():
    """(a) with param_wire='bf16' only the weights HipLinear / HipConv2d consume as bf16 operands may travel rounded — a
    matrix-shaped parameter a module reads in fp32 (an embedding table of a non-Hip layer) must use the fp32 chunk, or data-parallel
    replicas drift apart; (b) `_masters_stale` is raised by the sync STEP (train_data_list), so a captured optimizer step — which
    replays without running optimizer_step()'s Python — still makes save_model refuse rounded masters."""
    import types
    from hcp_diffusion_amd import trainer as T
    from hcp_diffusion_amd.layers import HipLinear
    m = torch.nn.Module()
    m.lin = HipLinear(8, 8)
    m.table = torch.nn.Embedding(16, 8)
    ids = T._bf16_consumed(m)
    assert ids == {id(m.lin.weight)}
    chunk_of = T._chunker(False, "bf16", unet_names=False, bf16_ids=ids)
    assert chunk_of("lin.weight", m.lin.weight) < T.FP32_WIRE
    assert chunk_of("lin.bias", m.lin.bias) >= T.FP32_WIRE
    assert chunk_of("table.weight", m.table.weight) >= T.FP32_WIRE              # a matrix, but consumed in fp32
    # (b) the flag is set where the step syncs, whatever runs the optimizer afterwards
    calls = []
    st = types.SimpleNamespace(shard=True, pwire=torch.zeros(1))
    fake = types.SimpleNamespace(_micro=0, accum=1, _overlap=False, use_graph=False, _sent=set(), _opt_graph=types.SimpleNamespace(replay=lambda: calls.append("replay")),
                                 _opt_graph_sent=(), _masters_stale=False, _states=lambda: [st], all_reduce=lambda: None,
                                 _run_all=lambda data_list, exchange=False: torch.zeros(1), optimizer_step=lambda **kw: calls.append("eager"))
    T.NativeTrainer.train_data_list(fake, [dict(latents=torch.zeros(1, 4, 8, 8))])
    assert calls == ["replay"] and fake._masters_stale is True
