"""GPU-only (tools library): sweep a hand-written list of GEMM shapes over tile / split-K / loader-ring choices with hipGraph-replay
timing and write tools/tune_extra_<tag>.json + tools/tune_loaders_<tag>.json in the formats tools/gen_gemm_table.py and
tools/gen_loader_table.py merge.  Round 3: the shapes the batched cross-attention K/V projections and the rank-16 self-attention split
introduced (nothing traced them when the tables were swept).

  python tools/tune_shapes.py r3"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())
from tune_gemm_common import timeit, rnd

# (mode, M, N, K, has_k2, launches per step): mode 0 = plain GEMM, 3 = fused-LoRA GEMM
SHAPES = [(0, 308, 512, 768, 0, 1), (0, 308, 24960, 1280, 0, 1), (0, 308, 512, 24960, 0, 1), (0, 308, 24960, 768, 0, 1),
          (0, 154, 2240, 2048, 0, 1), (0, 154, 166400, 4288, 0, 1), (0, 154, 2240, 166400, 0, 1),
          (3, 2048, 2560, 1280, 1, 70), (3, 2048, 1280, 2560, 1, 70)]
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
extra, loaders = [], []
for mode, M, N, Kd, k2, cnt in SHAPES:
    a, b = rnd(M, Kd), rnd(N, Kd)
    if mode == 3:
        l, e = rnd(32, Kd), rnd(N, 32)
        fn = lambda: K.gemm_lora(a, b, l, e)
    else:
        o = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
        fn = lambda: K.gemm(a, b, out=o)
    nk1 = Kd // 64
    L = K.lib()
    L.hcp_debug_set_gemm_config(-1); L.hcp_debug_set_gemm_loaders(0)
    heur = round(timeit(fn), 1)
    best = (heur, None, None)
    for cid in range(13):
        for s in (1, 2, 4, 8, 16):
            if (s > 1 and (nk1 // s < 4 or mode == 3)):
                continue
            L.hcp_debug_set_gemm_config(cid + 16 * s)
            try:
                t = round(timeit(fn), 1)
            except Exception:  # noqa: BLE001
                continue
            if t < best[0]:
                best = (t, cid, s)
    if best[1] is not None:
        extra.append(dict(mode=mode, M=M, N=N, K=Kd, has_k2=k2, stride=1, up=0, cfg=best[1], split=best[2], us=best[0], heur=heur, count=cnt))
    cur = best[0]
    bl = (cur, None, None, None)
    for st in (1, 3, 4):
        L.hcp_debug_set_gemm_loaders(st)
        for cid in (13, 14, 15):
            for s in (1, 2, 4, 8):
                if s > 1 and (nk1 // s < 4 or mode == 3):
                    continue
                L.hcp_debug_set_gemm_config(cid + 16 * s)
                try:
                    t = round(timeit(fn), 1)
                except Exception:  # noqa: BLE001
                    continue
                if t < bl[0]:
                    bl = (t, cid, s, st)
    if bl[1] is not None:
        loaders.append(dict(mode=mode, M=M, N=N, K=Kd, has_k2=k2, stride=1, up=0, cfg=bl[1], split=bl[2], stages=bl[3], us=bl[0], cur=cur, count=cnt))
    L.hcp_debug_set_gemm_config(-1); L.hcp_debug_set_gemm_loaders(-1)
    print(f"mode {mode} M{M} N{N} K{Kd}: heuristic {heur} us | best tile {best[1]}/s{best[2]} {best[0]} us | best with loaders {bl[1]}/s{bl[2]}/ring{bl[3]} {bl[0]} us", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(extra, open(os.path.join(ROOT, "gpurun_out", f"tune_extra_{tag}.json"), "w"), indent=0)
json.dump(loaders, open(os.path.join(ROOT, "gpurun_out", f"tune_loaders_{tag}.json"), "w"), indent=0)
