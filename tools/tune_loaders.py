"""GPU-only: where does the loader-wave variant of the v2 GEMM / implicit-conv kernel (csrc/gemm.hip, NLD = 4: four extra waves
issue the tile's LDS-DMA, the compute waves only read fragments and issue MFMAs) beat the dispatched configuration?

  python tools/tune_loaders.py [--pp] sd15|sdxl|dreambooth|controlnet [batch]

--pp (round 4): sweep the ping-pong kernel (csrc/gemm_pp.hip: two compute groups half a phase apart, K-split, LDS ring 3 / 4) on the
same tile ids against the FULL current dispatch (main + loader tables) -> gpurun_out/tune_pp_<workload>.json.

Traces one training step (kernels.TRACE), then for every distinct shape times the CURRENT dispatch against tile ids 13 / 14 / 15
(128x160, 64x160, 128x128 with 8 compute waves) x split-K with loaders forced on.  Winners by > 3 % go to
gpurun_out/tune_loaders_<workload>.json; tools/gen_loader_table.py turns tools/tune_loaders_*.json into csrc/gemm_tuned_loaders.inc
(looked up before the main table)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import autotune as A          # noqa: E402  (loads the tuning build, trace(), timeit(), rnd())
from hcp_diffusion_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
LOADER_CFGS = {13: "128x160w8+ld4", 14: "64x160w8+ld4", 15: "128x128w8+ld4"}


PP = "--pp" in sys.argv          # sweep the ping-pong kernel (csrc/gemm_pp.hip; loaders = 8 + ring) against the FULL current dispatch
if PP:
    sys.argv.remove("--pp")


def sweep_loaders(fn, nk1, cfgs=(13, 14, 15), splits=(1, 2, 4, 8)):
    K.lib().hcp_debug_set_gemm_loaders(-1 if PP else 0)
    K.lib().hcp_debug_set_gemm_config(-1)
    cur = round(A.timeit(fn), 1)
    res = {}
    for st in ((11, 12) if PP else (1, 3, 4)):  # LDS ring of 2 / 3 / 4 K tiles (ping-pong kernel: 8 + ring)
        K.lib().hcp_debug_set_gemm_loaders(st)
        for cid in cfgs:
            for s in splits:
                if s > 1 and nk1 // s < 4:
                    continue
                K.lib().hcp_debug_set_gemm_config(cid + 16 * s)
                try:
                    res[(cid, s, st)] = round(A.timeit(fn), 1)
                except Exception:  # noqa: BLE001
                    pass
    K.lib().hcp_debug_set_gemm_config(-1)
    K.lib().hcp_debug_set_gemm_loaders(-1)
    return cur, res


def main():
    workload = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else {"sdxl": 2, "dreambooth": 2}.get(workload, 4)
    keys = A.trace(workload, B)
    out, saved, total = [], 0.0, 0.0
    for key, cnt in sorted(keys.items(), key=lambda kv: -kv[1]):
        kind = key[0]
        if kind == "gemm":
            _, M, N, Kd, k2 = key
            if N % 4 or Kd % 64:
                continue
            a, b = A.rnd(M, Kd), A.rnd(N, Kd)
            a2, b2 = (A.rnd(M, 32), A.rnd(N, 32)) if k2 else (None, None)
            o = torch.empty(M, N, dtype=BF, device=dev)
            cur, res = sweep_loaders(lambda: K.gemm(a, b, a2=a2, b2=b2, out=o), Kd // 64)
            ent = dict(mode=0, M=M, N=N, K=Kd, has_k2=k2, stride=1, up=0)
        elif kind == "lora":
            _, M, N, Kd = key
            if Kd % 64:
                continue
            a, b, l, e = A.rnd(M, Kd), A.rnd(N, Kd), A.rnd(32, Kd), A.rnd(N, 32)
            cur, res = sweep_loaders(lambda: K.gemm_lora(a, b, l, e), Kd // 64, splits=(1,))
            ent = dict(mode=3, M=M, N=N, K=Kd, has_k2=1, stride=1, up=0)
        else:
            _, mode, Bn, Hs, Ws, C1, C2, cout, stride, up, Ho, Wo, ext = key
            if ext or (9 * (C1 + C2)) % 64 or C1 % 64 or C2 % 64:
                continue
            x1 = A.rnd(Bn, Hs, Ws, C1); x2 = A.rnd(Bn, Hs, Ws, C2) if C2 else None
            wp = A.rnd(cout, 3, 3, C1 + C2)
            fn = (lambda: K.conv3x3(x1, wp, cout, x2=x2, stride=stride, upsample=bool(up))) if mode == 0 else \
                 (lambda: K.conv3x3(x1, wp, cout, mode=1, stride=stride, out_hw=(Ho, Wo)))
            cur, res = sweep_loaders(fn, 9 * (C1 + C2) // 64)
            ent = dict(mode=1 if mode == 0 else 2, M=Bn * Ho * Wo, N=cout, K=9 * (C1 + C2), has_k2=0, stride=stride, up=up)
        total += cur * cnt
        if not res:
            continue
        us, (cid, s, st) = min((v, k) for k, v in res.items())
        ent.update(cfg=cid, split=s, stages=st, us=us, cur=cur, count=cnt, us_2stage=min(v for k, v in res.items() if k[2] == (11 if PP else 1)))
        out.append(ent)
        if us < 0.97 * cur:
            saved += (cur - us) * cnt
        print(f"{key} x{cnt}: dispatched {cur} us | loaders best {us} us ({LOADER_CFGS[cid]}/s{s}/ring{st}; 2-tile ring {ent['us_2stage']}){'  <-- wins' if us < 0.97 * cur else ''}", flush=True)
    print(f"GEMM-family time per step {total / 1e3:.2f} ms; loader variants would save {saved / 1e3:.2f} ms", flush=True)
    root = os.environ.get("GRAFT_REPO_ROOT", ROOT)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", f"tune_{'pp' if PP else 'loaders'}_{workload}.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
