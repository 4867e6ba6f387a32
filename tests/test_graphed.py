"""unet.enable_hip_graph() — the module-level graph path a trainer that is NOT NativeTrainer gets (the reference's Trainer.train_one_step,
train_ac.py:467-504; under `accelerate launch`: torch DDP around the model, train_ac.py:116-123,175) — for LoRA, for host-parameter
(DreamBooth.yaml:6-10) training and under stock DDP.  On the CPU the graphs are recorded callables (graphed._Recorded): the autograd
wiring, bucket bookkeeping, zero_grad semantics and the DDP hook firing are the same code; the GPU run of the same tests replays real
hipGraphs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


def _model(dev, mode):
    from hcp_diffusion_amd.lora import make_lora
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_
    torch.manual_seed(0)
    nat = NativeUNet2DConditionModel(**MICRO_CONFIG)
    nat.load_state_dict(seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1).state_dict())
    nat.to(dev)
    nat.requires_grad_(mode == "fullft")
    params = list(nat.parameters()) if mode == "fullft" else []
    if mode in ("lora", "both"):
        _, _, bucket = make_lora(nat, [dict(layers=PATS, rank=4)])
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for blk in bucket.blocks:
                blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g).to(dev) * 0.05)
        bucket.pack()
        params = [p for blk in bucket.blocks for p in (blk.layer.W_down, blk.layer.W_up)]
    if mode == "both":                                  # LoRA + a few host layers (the reference's `unet:` + `lora_unet:` lists together)
        host = [p for n, p in nat.named_parameters() if n.startswith("conv_in") or ".norm1." in n]
        for p in host:
            p.requires_grad_(True)
        params = params + host
    return nat, params


def _data(dev, n=2):
    g = torch.Generator().manual_seed(7)
    return [(torch.randn(2, 4, 8, 8, generator=g).to(dev), torch.randint(0, 1000, (2,), generator=g).to(dev),
             torch.randn(2, 9, 32, generator=g).to(torch.bfloat16).to(dev), torch.randn(2, 4, 8, 8, generator=g).to(dev)) for _ in range(n)]


def _loop(model, params, data, set_to_none, lr=1e-2):
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=1e-3)
    losses = []
    for x, t, ehs, target in data:
        pred = model(x, t, ehs).sample
        loss = torch.nn.functional.mse_loss(pred.float(), target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        opt.zero_grad(set_to_none=set_to_none)
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize("mode,set_to_none", [("lora", False), ("lora", True), ("fullft", True), ("both", True), ("fullft", False)])
def test_module_graph_trains_like_the_eager_module(backend, mode, set_to_none):
    """Two steps of an ordinary loop (module call, loss.backward(), clip_grad_norm_, torch AdamW, zero_grad) with the module replaying
    its forward / backward pair vs the same loop on the eager module: same losses, same parameters — for LoRA, for every host parameter
    (full fine-tune: the parameters are re-homed into a flat bucket at capture, the optimizer built BEFORE keeps working) and for both;
    zero_grad(set_to_none=True) (torch's default) must not leave stale sums in the buckets the captured kernels accumulate into."""
    dev = backend.device
    data = _data(dev)
    res = {}
    for graph in (False, True):
        nat, params = _model(dev, mode)
        if graph:
            nat.enable_hip_graph(True, _recorded_on_cpu=not backend.is_gpu)
        losses = _loop(nat, params, data, set_to_none)
        if graph:
            assert len(nat._hip_graphs) == 1                        # one signature, captured once
        res[graph] = (losses, torch.cat([p.detach().float().flatten().cpu() for p in params]))
    for a, b in zip(res[False][0], res[True][0]):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (res[False][0], res[True][0])
    pe, pg = res[False][1], res[True][1]
    assert ((pe - pg).norm() / pe.norm()).item() < 2e-3            # (LoRA: grouped vs per-layer weight-gradient launches round differently)
    assert res[True][0][0] != res[True][0][-1]


def test_module_graph_pending_forward_and_lru(backend):
    """A forward whose backward never runs (an evaluation pass in grad mode) must not park the signature on the eager path for good;
    signatures beyond MAX_SIGNATURES evict the least recently used pair."""
    from hcp_diffusion_amd import graphed
    dev = backend.device
    nat, params = _model(dev, "lora")
    nat.enable_hip_graph(True, _recorded_on_cpu=not backend.is_gpu)
    (x, t, ehs, target), = _data(dev, 1)
    y0 = nat(x, t, ehs).sample                       # never backpropagated
    del y0
    y1 = nat(x, t, ehs).sample
    assert type(y1.grad_fn).__name__ == "_GraphedFnBackward"       # replayed, not the eager fallback
    y1.float().square().mean().backward()
    old = graphed.MAX_SIGNATURES
    graphed.MAX_SIGNATURES = 2
    try:
        for h in (8, 12, 16):                        # three resolutions through a two-entry cache
            xs = torch.randn(1, 4, h, 8).to(dev)
            nat(xs, t[:1], ehs[:1]).sample.float().mean().backward()
        assert len(nat._hip_graphs) == 2
    finally:
        graphed.MAX_SIGNATURES = old


def _ddp_worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(emu_cdll())
    dev = torch.device("cpu")
    nat, params = _model(dev, mode)
    nat.enable_hip_graph(True, _recorded_on_cpu=True)
    ddp = torch.nn.parallel.DistributedDataParallel(nat, broadcast_buffers=False)      # train_ac.py:117
    data = [tuple(v[rank:rank + 1].contiguous() for v in d) for d in _data(dev)]
    _loop(ddp, params, data, set_to_none=True, lr=1e-2 if mode == "lora" else 1e-4)   # (every host weight moving by 1e-2 per step is chaos)
    assert len(nat._hip_graphs) == 1
    torch.save(torch.cat([p.detach().float().flatten() for p in params]), os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["lora", "fullft"])
def test_module_graph_under_stock_ddp(tmp_path, mode):
    """torch DDP around the graphed module, 2 ranks (gloo, interpreter): every trainable parameter is an input of the graph node, so the
    engine runs their AccumulateGrad nodes and the reducer's hooks average the bucket views after the backward pair was replayed.  Ranks
    end with identical parameters, equal (up to bf16 order) to one process stepping the eager module on both samples."""
    port = 32500 + os.getpid() % 2000
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0, r1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(emu_cdll())
    try:
        dev = torch.device("cpu")
        nat, params = _model(dev, mode)
        p_init = torch.cat([p.detach().float().flatten() for p in params])
        _loop(nat, params, _data(dev), set_to_none=True, lr=1e-2 if mode == "lora" else 1e-4)   # mse over the 2-sample batch = mean of the per-rank losses
        single = torch.cat([p.detach().float().flatten() for p in params])
        cos = torch.nn.functional.cosine_similarity(single - p_init, r0 - p_init, dim=0).item()
        assert cos > (0.97 if mode == "lora" else 0.9), cos         # three AdamW steps (sign flips of near-zero gradients aside)
        assert (r0 - p_init).abs().max().item() > (5e-3 if mode == "lora" else 1e-4)
    finally:
        K._set_backend_for_tests(None)


def test_a_second_signature_while_a_backward_is_pending_runs_eagerly_and_the_cap_evicts_the_oldest(backend):
    """All signatures of a module capture into ONE memory pool, so while ANY signature's forward awaits its backward no other signature
    may replay, capture or be evicted: fwd(shape A) -> fwd(shape B) -> backward of the sum must give the eager gradients.
    `enable_hip_graph(max_signatures=N)` (overlay key hip_graph_max_signatures) bounds the per-module cache; the least recently used
    pair goes first, with one warning."""
    import warnings
    dev = backend.device
    g = torch.Generator().manual_seed(11)
    xa, xb = torch.randn(2, 4, 8, 8, generator=g).to(dev), torch.randn(1, 4, 8, 8, generator=g).to(dev)
    ta, tb = torch.tensor([10, 500]).to(dev), torch.tensor([900]).to(dev)
    ea, eb = (torch.randn(n, 9, 32, generator=g).to(torch.bfloat16).to(dev) for n in (2, 1))
    grads = {}
    for graph in (False, True):
        nat, params = _model(dev, "lora")
        if graph:
            nat.enable_hip_graph(True, _recorded_on_cpu=not backend.is_gpu, max_signatures=2)
            for x, t, e in ((xa, ta, ea), (xb, tb, eb)):                    # both signatures captured, nothing pending
                nat(x, t, e).sample.float().sum().backward()
            for p in params:
                p.grad.zero_()
            assert len(nat._hip_graphs) == 2
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            la = nat(xa, ta, ea).sample.float().pow(2).mean()              # A replays: pending
            lb = nat(xb, tb, eb).sample.float().pow(2).mean()              # B must NOT replay into the shared pool: eager
            if graph:
                assert any("runs eagerly" in str(x.message) for x in w)
        (la + lb).backward()
        grads[graph] = torch.cat([p.grad.detach().float().flatten().cpu() for p in params])
        if graph:                                                           # a third signature: the cap drops the least recently used pair
            xc = torch.randn(1, 4, 16, 8, generator=g).to(dev)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                nat(xc, tb, eb).sample.float().sum().backward()
                assert any("input signatures" in str(x.message) for x in w)
            assert len(nat._hip_graphs) == 2
    cos = torch.nn.functional.cosine_similarity(grads[False], grads[True], dim=0).item()
    assert cos > 0.9999 and torch.allclose(grads[False], grads[True], rtol=2e-2, atol=1e-4), cos
