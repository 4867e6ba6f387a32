"""GPU-only lab: attention forward / backward of TWO builds of the library, interleaved in one process on the same operands.
usage: python tools/lab/attn_ab.py <other_build.so>      (the product build in the tree is "new", the argument is "old")
Rotating operand sets (cold L2), HIP events; also checks that both builds return the same bits."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import _lib
from hcp_diffusion_amd import kernels as K

BF = torch.bfloat16
dev = torch.device("cuda:0")
new = _lib.load()
old = _lib.bind(ctypes.CDLL(os.path.abspath(sys.argv[1])))
NSET = 8


def timeit(fn, iters=4 * NSET, warm=NSET):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, H = 4, 8
for (N, Nk, D, pre) in [(4096, 4096, 40, True), (4096, 4096, 40, False), (4096, 77, 40, True), (1024, 1024, 80, True), (1024, 77, 80, True),
                        (256, 256, 160, True), (4096, 4096, 64, True), (1024, 1024, 64, True)]:
    sets = [[torch.randn(B, n, H * D, device=dev).to(BF) for n in (N, Nk, Nk, N)] for _ in range(NSET)]
    res = {}
    for rnd in range(2):
        for name, lib in (("old", old), ("new", new)):
            K._set_backend_for_tests(lib)
            outs = [K.attention_fwd(q, k, v, H, q_prescaled=pre) for q, k, v, _ in sets]
            tf = timeit(lambda i: K.attention_fwd(*sets[i % NSET][:3], H, q_prescaled=pre))
            tb = timeit(lambda i: K.attention_bwd(*sets[i % NSET][:3], outs[i % NSET][0], sets[i % NSET][3], outs[i % NSET][1], H, q_prescaled=pre))
            g = K.attention_bwd(*sets[0][:3], outs[0][0], sets[0][3], outs[0][1], H, q_prescaled=pre)
            res.setdefault(name, []).append((tf, tb, [outs[0][0].clone()] + [t.clone() for t in g]))
    same = all(torch.equal(a, b) for a, b in zip(res["old"][0][2], res["new"][0][2]))
    fo, bo = min(r[0] for r in res["old"]), min(r[1] for r in res["old"])
    fn_, bn = min(r[0] for r in res["new"]), min(r[1] for r in res["new"])
    print(f"N{N} Nk{Nk} d{D} pre{int(pre)}: fwd old {fo:7.1f} new {fn_:7.1f} us | bwd old {bo:7.1f} new {bn:7.1f} us ({bn / bo - 1:+.1%}) | same bits: {same}", flush=True)
