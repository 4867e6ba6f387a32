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


def _bucket(tr):
    return tr.bucket if tr.bucket is not None else tr.host_buckets[0].bucket


def _make(tiny_cfg, mode="lora"):
    from hcp_diffusion_amd.trainer import NativeTrainer
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import OracleUNet2DConditionModel, seeded_init_
    torch.manual_seed(0)
    ora = seeded_init_(OracleUNet2DConditionModel(**tiny_cfg), 1)
    nat = NativeUNet2DConditionModel(**tiny_cfg)
    nat.load_state_dict(ora.state_dict())
    if mode == "fullft":                           # DreamBooth.yaml:6-10: every UNet parameter, one 3.4 GB-class bucket
        return NativeTrainer(nat, None, lr=1e-2, train_cfg=[dict(layers=[""])])
    tr = NativeTrainer(nat, [dict(layers=PATS, rank=4)], lr=1e-2)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in tr.bucket.blocks:
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
    tr.bucket.pack()
    return tr


def _data():
    g = torch.Generator().manual_seed(9)
    return (torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 24, 32, generator=g), torch.randn(2, 4, 8, 8, generator=g),
            torch.tensor([20, 700]))


def _worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    from oracle.unet_sd15 import MICRO_CONFIG as TINY_CONFIG      # two-level miniature: the DP contract does not need depth
    K._set_backend_for_tests(emu_cdll())
    tr = _make(TINY_CONFIG, mode)
    assert tr.world == world
    x0, ehs, noise, t = _data()
    sl = slice(rank, rank + 1)                     # rank r gets sample r of the global batch (strided sampler shard)
    tr.make_noise = lambda lat: (K.add_noise(lat, noise[sl], t[sl], tr.acp), noise[sl], t[sl])
    tr.forward_backward(x0[sl].contiguous(), ehs[sl].contiguous())
    tr.all_reduce()
    g = _bucket(tr).grads.clone() / world
    tr.optimizer_step()
    torch.save({"grads": g, "params": _bucket(tr).params.clone()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("mode", ["lora", "fullft"])
def test_two_rank_gloo_matches_single_process(tmp_path, mode):
    port = 29500 + os.getpid() % 2000 + (7 if mode == "fullft" else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["params"], r1["params"])
    # single process on the concatenated batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    from oracle.unet_sd15 import MICRO_CONFIG as TINY_CONFIG      # two-level miniature: the DP contract does not need depth
    K._set_backend_for_tests(emu_cdll())
    try:
        tr = _make(TINY_CONFIG, mode)
        x0, ehs, noise, t = _data()
        tr.make_noise = lambda lat: (K.add_noise(lat, noise, t, tr.acp), noise, t)
        tr.forward_backward(x0, ehs)
        g = _bucket(tr).grads.clone()
        tr.optimizer_step()
        cos = torch.nn.functional.cosine_similarity(g, r0["grads"], dim=0).item()
        assert cos > 0.9999, cos                                    # same math, different bf16 rounding order only
        assert ((g - r0["grads"]).norm() / g.norm()).item() < 1e-2
    finally:
        K._set_backend_for_tests(None)
