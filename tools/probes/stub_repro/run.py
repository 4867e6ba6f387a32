"""GPU probe (VERDICT r5 next #1c): can a __global__ TEMPLATE in a header (weak linkage: one host stub kept by the linker), compiled into several translation units of ONE shared library
built with -fvisibility=hidden (the round-5 form of hcp_fill32_kernel), corrupt atomics-based results — including results whose
accumulator was cleared by torch.zeros and never touched by the fill kernel (tests/test_kernels.py::test_lora_wgrad_and_pack, wrong by
0.70 in profiles/r5_gpu_tests_run_with_9_failures.txt)?
Builds libstub_shared.so (weak template linkage, 11 TUs) and libstub_static.so (-DSTUB_LINKAGE=static) with hipcc, then for each: 300 rounds of
{accumulator poisoned or zeroed by torch, optional fill through the header kernel from a rotating TU, 2048 x 256 integer atomic adds,
exact compare}, eagerly and as 200 replays of a captured hipGraph.   python tools/probes/stub_repro/run.py"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NTU = 11
N = 4099


def build(name, extra):
    objs = []
    for i in range(NTU):
        o = f"/tmp/stub_{name}_{i}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", f"-DTU={i}", *extra, "-c",
                        os.path.join(HERE, "tu.cpp"), "-o", o], check=True)
        objs.append(o)
    so = f"/tmp/libstub_{name}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "--offload-arch=gfx950", "-o", so, *objs], check=True)
    return ctypes.CDLL(so)


def expected():
    w = torch.arange(2048, dtype=torch.int64).view(-1, 1); t = torch.arange(256, dtype=torch.int64).view(1, -1)
    return torch.zeros(N, dtype=torch.int64).index_add_(0, ((37 * w + 101 * t) % N).flatten(), ((t & 3) + 1).expand(2048, 256).flatten()).double()


def main():
    dev = torch.device("cuda:0")
    want = expected()
    for name, extra in (("shared", []), ("static", ["-DSTUB_LINKAGE=static"])):
        lib = build(name, extra)
        fns = [getattr(lib, f"run_tu_{i}") for i in range(NTU)]
        for f in fns:
            f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; f.restype = ctypes.c_int
        st = torch.cuda.current_stream().cuda_stream
        bad = {"fill": 0, "torch_zeros": 0, "graph_fill": 0}
        for r in range(300):
            acc = torch.full((N,), float("nan"), device=dev)                       # poisoned: only the header's fill kernel clears it
            assert fns[r % NTU](acc.data_ptr(), N, 1, st) == 0
            bad["fill"] += int((acc.cpu().double() != want).any())
            acc2 = torch.zeros(N, device=dev)                                      # cleared by torch: the fill kernel is not involved at all
            assert fns[(r + 3) % NTU](acc2.data_ptr(), N, 0, st) == 0
            bad["torch_zeros"] += int((acc2.cpu().double() != want).any())
        acc = torch.full((N,), float("nan"), device=dev)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for i in range(NTU):                                               # every TU's fill + add in one captured chain (each fill re-clears)
                    assert fns[i](acc.data_ptr(), N, 1, side.cuda_stream) == 0
        for r in range(200):
            acc.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            bad["graph_fill"] += int((acc.cpu().double() != want).any())
        print(f"lib{name}: wrong results in {bad['fill']}/300 eager fill rounds, {bad['torch_zeros']}/300 torch.zeros rounds, {bad['graph_fill']}/200 graph replays", flush=True)


if __name__ == "__main__":
    sys.exit(main())
