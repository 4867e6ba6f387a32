"""Data-parallel path on CPU: 2 processes, gloo, kernels interpreted (tests/emu).  Checks the reference's DDP contract
(SURVEY.md §8e): after ONE all-reduce(SUM) of the flat LoRA gradient bucket and the 1/world factor folded into the
optimizer, both ranks hold identical parameters, equal to a single process stepping on the concatenated global batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


class _Both:
    """UNet + text-encoder LoRA buckets seen as one flat vector (mode 'lora_te')."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    grads = property(lambda self: torch.cat([self.a.grads, self.b.grads]))
    params = property(lambda self: torch.cat([self.a.params, self.b.params]))


def _bucket(tr):
    if getattr(tr, "te_bucket", None) is not None:
        return _Both(tr.bucket, tr.te_bucket)
    return tr.bucket if tr.bucket is not None else tr.host_buckets[0].bucket


def _make(tiny_cfg, mode="lora"):
    from hcp_diffusion_amd.trainer import NativeTrainer
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import OracleUNet2DConditionModel, seeded_init_
    torch.manual_seed(0)
    ora = seeded_init_(OracleUNet2DConditionModel(**tiny_cfg), 1)
    nat = NativeUNet2DConditionModel(**tiny_cfg)
    nat.load_state_dict(ora.state_dict())
    if mode in ("fullft", "fullft_sharded"):       # DreamBooth.yaml:6-10: every UNet parameter, one 3.4 GB-class bucket
        return NativeTrainer(nat, None, lr=1e-2, train_cfg=[dict(layers=[""])], shard_optimizer=(mode == "fullft_sharded"))
    te = None
    if mode == "lora_te":                          # lora_conventional.yaml as shipped: lora_unet + lora_text_encoder, two buckets
        from hcp_diffusion_amd.text_encoder import NativeCLIPTextModel
        from oracle.clip_ref import OracleCLIPTextModel
        tcfg = dict(vocab_size=100, hidden_size=tiny_cfg["cross_attention_dim"], intermediate_size=128, num_hidden_layers=2,
                    num_attention_heads=1, max_position_embeddings=77)
        te = NativeCLIPTextModel(**tcfg)
        te.load_state_dict(seeded_init_(OracleCLIPTextModel(**tcfg), 2).state_dict())
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-2, text_encoder=te,
                       lora_te_cfg=[dict(layers=[r"re:.*self_attn$", r"re:.*mlp$"], rank=4)] if te is not None else None)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for bk in [tr.bucket] + ([tr.te_bucket] if te is not None else []):
            for blk in bk.blocks:
                blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
            bk.pack()
    return tr


def _data():
    g = torch.Generator().manual_seed(9)
    return (torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 24, 32, generator=g), torch.randn(2, 4, 8, 8, generator=g),
            torch.tensor([20, 700]))


def _cfg(mode):
    from oracle.unet_sd15 import MICRO_CONFIG      # two-level miniature: the DP contract does not need depth
    return dict(MICRO_CONFIG, cross_attention_dim=64) if mode == "lora_te" else MICRO_CONFIG


def _step(tr, x0, ehs, mode, sl=slice(None)):
    if mode == "lora_te":                          # the prompt is encoded inside the step
        ids = torch.randint(0, 100, (2, 77), generator=torch.Generator().manual_seed(3))
        return tr.forward_backward(x0[sl].contiguous(), None, prompt_ids=ids[sl].contiguous())
    return tr.forward_backward(x0[sl].contiguous(), ehs[sl].contiguous())


def _worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(emu_cdll())
    tr = _make(_cfg(mode), mode)
    assert tr.world == world
    x0, ehs, noise, t = _data()
    sl = slice(rank, rank + 1)                     # rank r gets sample r of the global batch (strided sampler shard)
    tr.make_noise = lambda lat: (K.add_noise(lat, noise[sl], t[sl], tr.acp), noise[sl], t[sl])
    _step(tr, x0, ehs, mode, sl)
    tr.all_reduce()
    g = _bucket(tr).grads.clone() / world          # sharded buckets are exchanged inside optimizer_step: still the local gradient here
    if mode == "fullft_sharded":
        st = tr.host_buckets[0]
        assert st.shard and st.exp_avg.numel() * world == st.bucket.params.numel()      # moments exist for this rank's slice only
    tr.optimizer_step()
    torch.save({"grads": g, "params": _bucket(tr).params.clone()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["lora", "fullft", "fullft_sharded", "lora_te"])
def test_two_rank_gloo_matches_single_process(tmp_path, mode):
    """fullft_sharded: reduce-scatter -> AdamW on this rank's slice -> all-gather (the path full fine-tune / ControlNet buckets take
    under data parallelism) must leave both ranks with bit-identical parameters that equal the single-process step."""
    port = 29500 + os.getpid() % 2000 + {"lora": 0, "fullft": 7, "lora_te": 13, "fullft_sharded": 19}[mode]
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["params"], r1["params"])
    if mode == "fullft_sharded":                   # local (un-exchanged) gradients differ per rank; their mean is the global gradient
        r0["grads"] = r0["grads"] + r1["grads"]
    else:
        assert torch.equal(r0["grads"], r1["grads"])
    # single process on the concatenated batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(emu_cdll())
    try:
        tr = _make(_cfg(mode), mode)
        x0, ehs, noise, t = _data()
        tr.make_noise = lambda lat: (K.add_noise(lat, noise, t, tr.acp), noise, t)
        _step(tr, x0, ehs, mode)
        g = _bucket(tr).grads.clone()
        tr.optimizer_step()
        n = min(g.numel(), r0["grads"].numel())                     # (a sharded bucket is padded to a multiple of world * 64)
        g, r0["grads"] = g[:n], r0["grads"][:n]
        cos = torch.nn.functional.cosine_similarity(g, r0["grads"], dim=0).item()
        assert cos > 0.9999, cos                                    # same math, different bf16 rounding order only
        assert ((g - r0["grads"]).norm() / g.norm()).item() < 1e-2
        dp = (_bucket(tr).params[:n] - r0["params"][:n]).abs().max().item()
        assert dp < 2.5e-2, dp                                      # lr 1e-2: one AdamW step moves a parameter by <= lr; sign flips of tiny gradients aside
    finally:
        K._set_backend_for_tests(None)


def _ddp_model(tiny_cfg):
    """Frozen native UNet + LoRA the way the reference's trainer holds it: no NativeTrainer, gradients in `.grad`."""
    from hcp_diffusion_amd.lora import make_lora
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import OracleUNet2DConditionModel, seeded_init_
    torch.manual_seed(0)
    nat = NativeUNet2DConditionModel(**tiny_cfg)
    nat.load_state_dict(seeded_init_(OracleUNet2DConditionModel(**tiny_cfg), 1).state_dict())
    nat.requires_grad_(False)
    _, _, bucket = make_lora(nat, [dict(layers=PATS, rank=4)])
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in bucket.blocks:
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
    bucket.pack()
    return nat, bucket


def _plain_steps(model, unet_params, x, ehs, t, target, steps=2):
    opt = torch.optim.AdamW(unet_params, lr=1e-2, weight_decay=1e-3)
    for _ in range(steps):
        pred = model(x, t, ehs).sample
        loss = torch.nn.functional.mse_loss(pred.float(), target)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=False)


def _ddp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    from oracle.unet_sd15 import MICRO_CONFIG
    K._set_backend_for_tests(emu_cdll())
    nat, bucket = _ddp_model(MICRO_CONFIG)
    ddp = torch.nn.parallel.DistributedDataParallel(nat, broadcast_buffers=False)       # train_ac.py:117: DistributedDataParallelKwargs(broadcast_buffers=False)
    x0, ehs, noise, t = _data()
    sl = slice(rank, rank + 1)
    params = [p for blk in bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
    _plain_steps(ddp, params, x0[sl].contiguous(), ehs[sl].to(torch.bfloat16).contiguous(), t[sl], noise[sl].contiguous())
    torch.save({"params": bucket.params.clone()}, os.path.join(out, f"ddp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_stock_torch_ddp_around_the_native_unet(tmp_path):
    """The reference's multi-GPU loop wraps the model in torch DDP (accelerate, train_ac.py:117-123,175).  The native layers write
    parameter gradients in place and return None to autograd; the engine still runs each parameter's AccumulateGrad node (with an
    undefined gradient), so the reducer's hooks fire AFTER the kernel that wrote `.grad` was enqueued and stock DDP averages the
    bucket views like any other gradient: two steps, both ranks end with identical parameters, equal to one process on both samples."""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "ddp0.pt"); r1 = torch.load(tmp_path / "ddp1.pt")
    assert torch.equal(r0["params"], r1["params"])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    from oracle.unet_sd15 import MICRO_CONFIG
    K._set_backend_for_tests(emu_cdll())
    try:
        nat, bucket = _ddp_model(MICRO_CONFIG)
        x0, ehs, noise, t = _data()
        params = [p for blk in bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
        _plain_steps(nat, params, x0, ehs.to(torch.bfloat16), t, noise)          # mse over the 2-sample batch = mean of the per-rank losses
        p_init = _ddp_model(MICRO_CONFIG)[1].params
        upd_single, upd_ddp = bucket.params - p_init, r0["params"] - p_init
        cos = torch.nn.functional.cosine_similarity(upd_single, upd_ddp, dim=0).item()
        assert cos > 0.98, cos                                                  # same two AdamW updates (sign flips of near-zero gradients aside)
        assert (upd_ddp.abs().max().item() > 5e-3)                              # ... and they are real updates
    finally:
        K._set_backend_for_tests(None)


# ---- sharded exchange variants: chunks reduce-scattered from backward, bf16 on the wire (trainer.py overlap_exchange / *_wire)
VARIANTS = {"overlap": dict(overlap_exchange=True),
            "overlap_bf16_grads": dict(overlap_exchange=True, grad_wire="bf16"),
            "bf16_both": dict(overlap_exchange=True, grad_wire="bf16", param_wire="bf16"),
            "bf16_params_no_overlap": dict(param_wire="bf16")}


def _fullft(tiny_cfg, **kw):
    from hcp_diffusion_amd.trainer import NativeTrainer
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import OracleUNet2DConditionModel, seeded_init_
    nat = NativeUNet2DConditionModel(**tiny_cfg)
    nat.load_state_dict(seeded_init_(OracleUNet2DConditionModel(**tiny_cfg), 1).state_dict())
    return NativeTrainer(nat, None, lr=1e-3, train_cfg=[dict(layers=[""])], shard_optimizer=True, **kw)


def _variant_worker(rank, world, port, out, variant):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    from oracle.unet_sd15 import MICRO_CONFIG
    K._set_backend_for_tests(emu_cdll())
    x0, ehs, noise, t = _data()
    sl = slice(rank % 2, rank % 2 + 1)             # (two distinct samples; worlds of 4 / 8 see each of them on half of the ranks)
    res = {}
    for name, kw in (("base", {}), ("var", VARIANTS[variant])):
        tr = _fullft(MICRO_CONFIG, **kw)
        st = tr.host_buckets[0]
        assert st.shard and len(st.parts) == (3 if kw.get("overlap_exchange") else 1) + (kw.get("param_wire") == "bf16")
        tr.make_noise = lambda lat: (K.add_noise(lat, noise[sl], t[sl], tr.acp), noise[sl], t[sl])
        sent = []
        for _ in range(2):                          # two steps: the second one sees gradients cleared by the first (fused into the wire cast)
            tr.train_one_step(x0[sl].contiguous(), ehs[sl].contiguous())
            sent.append(sorted(tr._sent))
        res[name] = {n: p.detach().clone() for n, p in tr.unet.named_parameters()}
        if name == "var":
            assert sent == [[0, 1], [0, 1]] if kw.get("overlap_exchange") else sent == [[], []]
            assert st.bucket.grads.abs().max().item() == 0
            res["var_before_sync"] = res["var"]
            if kw.get("param_wire") == "bf16":             # save_model must not write rounded masters, nor issue a collective on its own
                import pytest as _pt
                with _pt.raises(RuntimeError, match="sync_masters"):
                    tr.save_model(None, 0)
            tr.sync_masters()
            res["var"] = {n: p.detach().clone() for n, p in tr.unet.named_parameters()}
            res["own"] = [(lo + rank * own, lo + (rank + 1) * own) for _, lo, hi, own, _ in st.parts]
    torch.save(res, os.path.join(out, f"v{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_sharded_exchange_variants(tmp_path, variant):
    """Full fine-tune on 2 ranks (gloo, interpreter), two optimisation steps, against the plain sharded path (one reduce-scatter after
    backward, fp32 both ways):
    * overlap: the three chunks leave from backward hooks in completion order — same sums, same parameters (the clip norm adds the slices
      in another order: 1e-6);
    * bf16 gradients on the wire: torch DDP's bf16_compress_hook numerics — AdamW updates agree to the rounding of the gradients;
    * bf16 parameters on the wire: every rank derives BIT-IDENTICAL bf16 operands; the fp32 masters are exact on the owner and, after
      sync_masters(), everywhere."""
    port = 30100 + os.getpid() % 2000 + list(VARIANTS).index(variant) * 3
    mp.spawn(_variant_worker, args=(2, port, str(tmp_path), variant), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "v0.pt"), torch.load(tmp_path / "v1.pt")
    kw = VARIANTS[variant]
    flat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    base, var0, var1 = flat(r0["base"]), flat(r0["var"]), flat(r1["var"])
    assert torch.equal(flat(r0["base"]), flat(r1["base"]))
    assert torch.equal(var0, var1)                                  # masters agree on both ranks (after the sync where it is needed)
    b0, b1 = flat(r0["var_before_sync"]), flat(r1["var_before_sync"])
    assert torch.equal(b0.to(torch.bfloat16), b1.to(torch.bfloat16))       # what the layers compute with is the same everywhere, always
    vec = lambda d: torch.cat([d[n].flatten() for n in sorted(d) if d[n].dim() <= 1])
    assert torch.equal(vec(r0["var_before_sync"]), vec(r1["var_before_sync"]))   # biases / norm affine (fp32 in the kernels): never rounded
    if kw.get("param_wire") == "bf16":
        assert not torch.equal(b0, b1)                                       # (the fp32 tails of a slice live on its owner only)
    else:
        assert torch.equal(b0, b1)
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_
    init = flat({n: p.detach() for n, p in seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1).named_parameters()})
    moved = (base - init).norm().item()
    assert moved > 0
    err = (var0 - base).norm().item() / moved
    assert err < (1e-4 if kw.get("grad_wire") != "bf16" else 5e-2), err


@pytest.mark.slow
@pytest.mark.parametrize("world,variant", [(4, "bf16_both"), (8, "overlap")])
def test_sharded_exchange_worlds_of_4_and_8(tmp_path, world, variant):
    """the sharded exchange beyond two ranks, on gloo + the interpreter — chunk padding to world * 64 elements, slice
    alignment of the 16-byte AdamW accesses (own = padded / world), the three-chunk early exchange and both bf16 wires with 4 and 8
    owners: every rank ends with the same fp32 masters (after sync_masters() where the parameter wire is bf16), every rank computes with
    the same bf16 operands before it, and the result agrees with the plain one-reduce-scatter fp32 path on the same world."""
    port = 31200 + os.getpid() % 2000 + world * 3 + list(VARIANTS).index(variant)
    mp.spawn(_variant_worker, args=(world, port, str(tmp_path), variant), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"v{r}.pt") for r in range(world)]
    kw = VARIANTS[variant]
    flat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    base = flat(rs[0]["base"])
    for r in rs[1:]:
        assert torch.equal(flat(r["base"]), base) and torch.equal(flat(r["var"]), flat(rs[0]["var"]))
        assert torch.equal(flat(r["var_before_sync"]).to(torch.bfloat16), flat(rs[0]["var_before_sync"]).to(torch.bfloat16))
    # the slices the ranks own tile each part without gaps or overlaps, and start on 16-byte boundaries (4 fp32 = 16 B, 8 bf16 = 16 B)
    for part in range(len(rs[0]["own"])):
        spans = sorted(r["own"][part] for r in rs)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)) and all((lo % 8 == 0 and (hi - lo) % 64 == 0) for lo, hi in spans)
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_
    init = flat({n: p.detach() for n, p in seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1).named_parameters()})
    moved = (base - init).norm().item()
    err = (flat(rs[0]["var"]) - base).norm().item() / moved
    assert moved > 0 and err < (1e-4 if kw.get("grad_wire") != "bf16" else 5e-2), err
