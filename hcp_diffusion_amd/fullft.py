"""Full fine-tuning of host parameters (reference ``cfgs/train/examples/DreamBooth.yaml:6-10``:
``unet: [{lr: 1e-6, layers: ['']}]`` — every UNet parameter trainable; selection semantics of
``hcpdiff/train_ac.py:280-312`` / ``utils/net_utils.py get_match_layers``).

``HostBucket`` re-homes the selected fp32 master parameters as views of ONE flat buffer and gives each a ``.grad`` view
of ONE flat gradient buffer, so that — exactly as for the LoRA bucket — the DP exchange is a single all-reduce, clip +
AdamW a single fused kernel, and the refresh of the layers' bf16 operand copies a single grouped launch
(``csrc/pack.hip``).  3x3 convolution weights keep their diffusers shape [Cout,Cin,3,3] but are stored channels_last,
which IS the kernels' [Cout][ky][kx][Cin]: the weight-gradient kernel writes, and the pack kernel reads, that layout
directly (state_dict / checkpoints are unaffected: they go by logical shape).
"""
import numpy as np
import torch

from . import kernels as K
from . import ops
from .layers import HipConv2d, HipLinear

PIECE_DTYPE = np.dtype([("src", "<u8"), ("dst_rm", "<u8"), ("dst_tr", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("src_ld", "<i4"),
                        ("rm_ld", "<i4"), ("tr_ld", "<i4"), ("tile0", "<i4"), ("tiles_c", "<i4"), ("scale", "<f4")])


def _is_conv3(p):
    return p.dim() == 4 and p.shape[2] == 3 and p.shape[3] == 3


class HostBucket:
    def __init__(self, model, params, pad_multiple=1, chunk_of=None):
        """params: list of (name, parameter) to train (all on one device, fp32); pad_multiple: the length of the flat buffers (of
        every chunk of them) is rounded up to a multiple of it (sharded optimizer: equal slices per rank; the tails stay zero);
        chunk_of: (name, parameter) -> chunk id.  Parameters are laid out chunk by chunk (`self.chunks` = [(lo, hi)]): the trainer orders
        the chunks by WHEN backward finishes their gradients, so that a chunk's reduce-scatter can leave while the rest of
        backward still runs (trainer.py, `overlap_exchange`)."""
        assert params, "no trainable host parameter selected"
        dev = params[0][1].device
        if chunk_of is not None:
            params = sorted(params, key=lambda np_: chunk_of(*np_))          # stable: registration order inside a chunk
        offs, n, self.chunks, cur, lo = [], 0, [], None, 0       # chunks: [(chunk id, lo, hi)]
        for name, p in params:
            if p.dtype != torch.float32:
                raise TypeError("hcp_diffusion_amd: full fine-tuning keeps fp32 master parameters (reference: fp32 params + autocast)")
            c = chunk_of(name, p) if chunk_of is not None else 2          # (2 = 'complete when backward ends')
            if cur is not None and c != cur:
                n = (n + pad_multiple - 1) // pad_multiple * pad_multiple
                self.chunks.append((cur, lo, n)); lo = n
            cur = c
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4                      # 16-byte aligned segments
        n = (n + pad_multiple - 1) // pad_multiple * pad_multiple
        self.chunks.append((cur, lo, n))
        self.numel = n
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.named = list(params)
        self._gv = []                                   # (parameter, its .grad view): re-attached after zero_grad(set_to_none=True)
        with torch.no_grad():
            for (name, p), o in zip(params, offs):
                seg, gseg = self.params[o:o + p.numel()], self.grads[o:o + p.numel()]
                if _is_conv3(p):
                    co, ci = p.shape[0], p.shape[1]
                    v = seg.view(co, 3, 3, ci).permute(0, 3, 1, 2)
                    g = gseg.view(co, 3, 3, ci).permute(0, 3, 1, 2)
                else:
                    v, g = seg.view(p.shape), gseg.view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = g
                p.requires_grad_(True)
                p._hcp_bucket = self
                self._gv.append((p, g))
        owned = {id(p) for _, p in params}
        self.layers = [m for m in model.modules() if isinstance(m, (HipLinear, HipConv2d)) and id(m.weight) in owned]
        self._pieces = None
        self._packed_ver = None

    # ---- a trainer that is not NativeTrainer steps / zeroes through the Parameters (graphed.py)
    def _version(self):
        return sum(p._version for p, _ in self._gv)

    def stale(self):
        """Has an optimizer written the fp32 masters since the bf16 operands were derived?  (every in-place step bumps p._version)"""
        return self._packed_ver != self._version()

    def attach_grads(self):
        for p, g in self._gv:
            if p.grad is None:
                p.grad = g

    def zero_dropped(self):
        """zero_grad(set_to_none=True) dropped `.grad` views = the trainer's zero: clear those slices, hand the views back."""
        dropped = [(p, g) for p, g in self._gv if p.grad is None]
        if dropped and len(dropped) == len(self._gv):
            self.grads.zero_()
        for p, g in dropped:
            if len(dropped) != len(self._gv):
                g.zero_()
            p.grad = g

    # ---- bf16 operand refresh
    def _build_pieces(self):
        rows = []
        tile0 = 0
        self._pks = []
        for m in self.layers:
            pk = m.packed()                                     # (re)built against the re-homed parameter storage
            self._pks.append(pk)
            w = m.weight
            if isinstance(m, HipConv2d) and m.kernel_size == (3, 3):
                co, ci = w.shape[0], w.shape[1]
                src = w.permute(0, 2, 3, 1)
                assert src.is_contiguous()
                for tap in range(9):
                    tc = (ci + 63) // 64
                    rows.append((src.data_ptr() + 4 * tap * ci, pk.w.data_ptr() + 2 * tap * pk.cin_pad,
                                 pk.wd.data_ptr() + 2 * tap * pk.cout_pad, co, ci, 9 * ci, 9 * pk.cin_pad, 9 * pk.cout_pad, tile0, tc, 1.0))
                    tile0 += ((co + 63) // 64) * tc
            else:
                nn_, kk = w.shape[0], w.numel() // w.shape[0]
                assert w.is_contiguous()
                tc = (kk + 63) // 64
                rows.append((w.data_ptr(), pk.w.data_ptr(), pk.wt.data_ptr(), nn_, kk, kk, kk, nn_, tile0, tc, 1.0))
                tile0 += ((nn_ + 63) // 64) * tc
        arr = np.array(rows, dtype=PIECE_DTYPE)
        assert arr.dtype.itemsize == K.lib().hcp_pack_piece_bytes()
        self._pieces = torch.from_numpy(arr.view(np.uint8).copy()).to(self.params.device)
        self._n_pieces, self._tiles = len(rows), tile0

    def repack(self):
        """bf16 forward / data-gradient operands of every trained layer <- fp32 masters, in place, one launch."""
        if self._pieces is None:
            self._build_pieces()
        elif not torch.cuda.is_current_stream_capturing() if self.params.is_cuda else True:
            for m, pk in zip(self.layers, self._pks):           # someone re-packed a layer behind our back (in-place torch
                if m._pk is not pk:                             # op on a parameter): re-derive the descriptor table
                    self._build_pieces()
                    break
        if self._n_pieces:
            K.pack_weights(self._pieces, self._n_pieces, self._tiles)
        ops.invalidate_merged_cache()
        self._packed_ver = self._version()
