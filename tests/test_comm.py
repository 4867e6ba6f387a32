"""The C-ABI communicator (csrc/comm.hip): argument checks without a GPU; on the GPU a one-rank RCCL communicator runs the
three flat collectives eagerly and inside a captured hipGraph (the 8-GPU run itself is the driver's)."""
import ctypes

import pytest
import torch

from hcp_diffusion_amd import _lib


def test_comm_entry_points_reject_bad_arguments():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.hcp_comm_init(0, 1, None, ctypes.byref(h)) < 0 and b"null" in lib.hcp_last_error()
    assert lib.hcp_comm_init(3, 2, ctypes.create_string_buffer(128), ctypes.byref(h)) < 0 and b"rank 3 of 2" in lib.hcp_last_error()
    assert lib.hcp_allreduce_flat(None, None, 4, 0, None) < 0
    assert lib.hcp_reduce_scatter_flat(None, None, None, 4, 0, None) < 0
    assert lib.hcp_allgather_flat(None, None, None, 4, 0, None) < 0
    assert lib.hcp_comm_rank(None) == -1 and lib.hcp_comm_world(None) == -1
    assert lib.hcp_comm_destroy(None) == 0


def test_null_and_torch_comm_single_process():
    from hcp_diffusion_amd.comm import NullComm, make_comm
    c = make_comm(torch.device("cpu"))
    assert isinstance(c, NullComm) and c.world == 1
    a = torch.arange(8.0); r = torch.zeros(8)
    c.reduce_scatter(a, r); assert torch.equal(a, r)
    c.all_gather(a, r); assert torch.equal(a, r)
    assert c.all_reduce_(a) is a


@pytest.mark.gpu
def test_abi_comm_one_rank_rccl_eager_and_captured():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from hcp_diffusion_amd import kernels as K
    from hcp_diffusion_amd.comm import AbiComm
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    c = AbiComm(0, 1, AbiComm.new_unique_id(), dev)                  # dlopen RCCL, ncclCommInitRank over one rank
    assert K.lib().hcp_comm_world(c._h) == 1 and K.lib().hcp_comm_rank(c._h) == 0
    x = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    y = x.clone()
    c.all_reduce_(y)
    r = torch.empty_like(x); g = torch.zeros_like(x)
    c.reduce_scatter(x, r); c.all_gather(x, g)
    xb = x.to(torch.bfloat16); yb = xb.clone(); c.all_reduce_(yb)
    torch.cuda.synchronize()
    assert torch.equal(x, y) and torch.equal(x, r) and torch.equal(x, g) and torch.equal(xb, yb)
    # stream-ordered, no allocation, no sync: capturable like a kernel launch
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c.all_reduce_(y)                                             # warm the communicator on this stream
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    z = torch.full((4096,), 3.0, device=dev)
    with torch.cuda.graph(gr):
        c.all_reduce_(z)
        z.mul_(2.0)
    gr.replay(); gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(z, torch.full((4096,), 12.0, device=dev))
    c.close()


@pytest.mark.gpu
def test_sharded_overlapped_exchange_through_rccl_inside_the_captured_step():
    """The full-fine-tune exchange with everything on (chunks reduce-scattered from backward hooks on the side stream, bf16 wires), the
    collectives going through RCCL behind the C ABI (one rank: ncclReduceScatter / ncclAllGather degenerate to copies but are the real
    calls), the forward/backward + early reduce-scatters CAPTURED as one hipGraph with a side branch: equals the plain eager trainer."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from hcp_diffusion_amd import kernels as K
    from hcp_diffusion_amd.comm import AbiComm
    from hcp_diffusion_amd.trainer import NativeTrainer
    from test_trainer import _batch, _fix_noise, _native
    K._set_backend_for_tests(None)
    dev = torch.device("cuda:0")
    c = AbiComm(0, 1, AbiComm.new_unique_id(), dev)
    data = [dict(**_batch(dev, 1)), dict(**_batch(dev, 2), loss_weight=0.5)]
    res = []
    for kw in ({}, dict(comm=c, shard_optimizer="force", overlap_exchange=True, grad_wire="bf16", param_wire="bf16", use_graph=True)):
        tr = NativeTrainer(_native(dev), None, lr=1e-3, train_cfg=[dict(layers=[""])], **kw)
        _fix_noise(tr, dev)
        for _ in range(3):
            tr.train_data_list([dict(d) for d in data])
        torch.cuda.synchronize()
        res.append(torch.cat([p.detach().float().flatten().cpu() for _, p in sorted(tr.unet.named_parameters())]))
    init = torch.cat([p.detach().float().flatten() for _, p in sorted(_native("cpu").named_parameters())])
    moved = (res[0] - init).norm().item()
    assert moved > 0 and (res[1] - res[0]).norm().item() / moved < 5e-2
    c.close()
