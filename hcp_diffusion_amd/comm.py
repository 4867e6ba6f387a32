"""The data-parallel exchange of the step (reference: accelerate -> torch DDP, train_ac.py:117-123,175,482).

Two interchangeable back ends with the same three flat-buffer collectives:

* ``AbiComm``   — RCCL through the C ABI (``hcp_comm_*`` / ``hcp_allreduce_flat`` / ``hcp_reduce_scatter_flat`` /
  ``hcp_allgather_flat``, csrc/comm.hip): raw device pointers on the caller's HIP stream, graph-capturable.  The 128-byte
  rendezvous token is created by rank 0 and handed to the other ranks through the ``torch.distributed`` store (any
  key-value side channel would do: this is the only use of torch.distributed on that path).
* ``TorchComm`` — ``torch.distributed`` collectives (backend "nccl" = RCCL on ROCm, or "gloo" on the CPU for the 2-rank
  tests that run the interpreted kernels).

``make_comm()`` picks: HCP_COMM=abi|torch overrides; default = torch.distributed when a process group exists.
"""
import ctypes
import os

import torch

from . import _lib
from . import kernels as K

_DT = {torch.float32: 0, torch.bfloat16: 1}


class NullComm:
    """Single process: nothing to exchange."""
    world, rank = 1, 0

    def all_reduce_(self, t):
        return t

    def reduce_scatter(self, send, recv):
        if recv.data_ptr() != send.data_ptr():
            recv.copy_(send[:recv.numel()])
        return recv

    def all_gather(self, send, recv):
        if recv.data_ptr() != send.data_ptr():
            recv[:send.numel()].copy_(send)
        return recv

    def barrier(self):
        pass


class TorchComm:
    def __init__(self, group=None):
        import torch.distributed as dist
        self._d, self.pg = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_reduce_(self, t):
        self._d.all_reduce(t, op=self._d.ReduceOp.SUM, group=self.pg)
        return t

    def reduce_scatter(self, send, recv):
        """recv[n] <- sum over ranks of send[rank*n:(rank+1)*n]."""
        n = recv.numel()
        assert send.numel() == n * self.world
        if self._d.get_backend(self.pg) == "gloo":             # gloo has no reduce_scatter: all-reduce, keep own slice
            tmp = send.clone()
            self._d.all_reduce(tmp, op=self._d.ReduceOp.SUM, group=self.pg)
            recv.copy_(tmp[self.rank * n:(self.rank + 1) * n])
        else:
            self._d.reduce_scatter_tensor(recv, send, op=self._d.ReduceOp.SUM, group=self.pg)
        return recv

    def all_gather(self, send, recv):
        """recv[r*n:(r+1)*n] <- rank r's send[n]."""
        n = send.numel()
        assert recv.numel() == n * self.world
        if self._d.get_backend(self.pg) == "gloo":
            parts = [torch.empty_like(send) for _ in range(self.world)]
            self._d.all_gather(parts, send.contiguous(), group=self.pg)
            for r, p_ in enumerate(parts):
                recv[r * n:(r + 1) * n].copy_(p_)
        else:
            self._d.all_gather_into_tensor(recv, send, group=self.pg)
        return recv

    def barrier(self):
        self._d.barrier(group=self.pg)


class AbiComm:
    """RCCL communicator behind the C ABI; collectives are enqueued on torch's current stream of the tensor's device."""

    def __init__(self, rank, world, unique_id: bytes, device):
        assert len(unique_id) == 128
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(K.lib().hcp_comm_init(rank, world, ctypes.c_char_p(unique_id), ctypes.byref(self._h)), "hcp_comm_init")

    @staticmethod
    def new_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(K.lib().hcp_comm_unique_id(buf), "hcp_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_store(cls, device, group=None, key="hcp_comm_uid"):
        """Bootstrap over an initialised torch.distributed group: rank 0 publishes the token via broadcast_object_list."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        err = None
        try:
            box = [cls.new_unique_id() if rank == 0 else None]
        except RuntimeError as e_:                    # librccl.so missing / without a symbol on rank 0: tell everybody instead of leaving them waiting
            box, err = [None], e_
        dist.broadcast_object_list(box, src=0, group=group)
        comm = None
        if box[0] is not None:
            try:
                comm = cls(rank, world, box[0], device)
            except RuntimeError as e_:
                err = e_
        # every rank learns whether EVERY rank has a communicator (one small all-reduce on the bootstrap group, which has its own
        # timeout): a partial failure must not leave the healthy ranks inside their first collective forever
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32,
                          device=device if dist.get_backend(group) != "gloo" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            return comm
        if comm is not None:
            comm.close()
        import warnings
        warnings.warn(f"hcp_diffusion_amd: RCCL through the C ABI could not be initialised on every rank ({err or 'another rank failed'}); "
                      "falling back to torch.distributed collectives for the gradient exchange", RuntimeWarning)
        return TorchComm(group)

    def _s(self, t):
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT, "AbiComm: contiguous fp32/bf16 device tensors"
        return torch.cuda.current_stream(t.device).cuda_stream

    def all_reduce_(self, t):
        _lib.check(K.lib().hcp_allreduce_flat(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], self._s(t)), "hcp_allreduce_flat")
        return t

    def reduce_scatter(self, send, recv):
        assert send.numel() == recv.numel() * self.world and send.dtype == recv.dtype
        _lib.check(K.lib().hcp_reduce_scatter_flat(self._h, send.data_ptr(), recv.data_ptr(), recv.numel(), _DT[recv.dtype], self._s(recv)),
                   "hcp_reduce_scatter_flat")
        return recv

    def all_gather(self, send, recv):
        assert recv.numel() == send.numel() * self.world and send.dtype == recv.dtype
        _lib.check(K.lib().hcp_allgather_flat(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), _DT[send.dtype], self._s(send)),
                   "hcp_allgather_flat")
        return recv

    def barrier(self):
        t = torch.zeros(1, device=self.device)
        self.all_reduce_(t)
        torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        if self._h:
            _lib.check(K.lib().hcp_comm_destroy(self._h), "hcp_comm_destroy")
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def make_comm(device, group=None, kind=None):
    """The step's communicator: NullComm without a process group; else HCP_COMM / `kind` = 'abi' | 'torch' (default torch)."""
    import torch.distributed as dist
    if not (group is not None or (dist.is_available() and dist.is_initialized())):
        return NullComm()
    if dist.get_world_size(group) == 1 and kind is None and "HCP_COMM" not in os.environ:
        return NullComm()
    kind = kind or os.environ.get("HCP_COMM", "torch")
    if kind == "abi":
        return AbiComm.from_torch_store(device, group)
    if kind == "torch":
        return TorchComm(group)
    raise ValueError(f"HCP_COMM={kind!r}: expected 'abi' or 'torch'")
