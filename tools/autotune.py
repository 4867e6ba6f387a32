"""GPU-only: tune the (tile, split-K) dispatch for the shapes a workload ACTUALLY launches.

  python tools/autotune.py sdxl|dreambooth|controlnet|sd15 [batch]

One eager training step runs with kernels.TRACE recording every GEMM / fused-LoRA GEMM / implicit-conv launch; every distinct
shape is then swept over all tile configurations and split-K factors through the tuning hook, and the winners are written to
gpurun_out/tune_extra_<workload>.json.  tools/gen_gemm_table.py merges tools/tune_extra_*.json into csrc/gemm_tuned.inc."""
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())      # tuning build: the hcp_debug_* hooks do not exist in the product library
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import SDXL_CONFIG, NativeUNet2DConditionModel

BF = torch.bfloat16
dev = torch.device("cuda:0")
CFG_NAMES = ["128x128", "128x64", "64x64", "128x160", "64x160", "256x128", "256x160", "128x320", "128x160w8s3", "128x160w4s3", "256x160w8s3",
             "128x320w16", "256x160w16", "128x160w8", "64x160w8", "128x128w8"]
PATS = [r"re:.*\.attn.?$", r"re:.*\.ff$"]


def timeit(fn, iters=10, warm=2):
    """us per call, measured on REPLAYS of a hipGraph holding `iters` calls: the Python wrapper + launch path costs 10-25 us per
    call, so a plain loop times the CPU, not the GPU, for every kernel shorter than that (half the launches of a training step)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del g
    return e0.elapsed_time(e1) / (iters * reps) * 1e3   # us


def rnd(*s):
    return torch.randn(*s, device=dev).to(BF)


def trace(workload, B):
    sdxl = workload == "sdxl"
    with torch.device("meta"):
        unet = NativeUNet2DConditionModel(**SDXL_CONFIG) if sdxl else NativeUNet2DConditionModel()
    unet = unet.to_empty(device=dev)
    with torch.no_grad():
        for n, p in unet.named_parameters():
            p.normal_(0, 0.02) if p.dim() > 1 else p.fill_(1.0 if n.endswith("weight") else 0.0)
    kw = {}
    if workload == "dreambooth":
        tr = NativeTrainer(unet, None, train_cfg=[dict(layers=[""], lr=1e-6)])
    elif workload == "controlnet":
        from hcp_diffusion_amd.controlnet import make_controlnet
        plug = make_controlnet(unet)
        with torch.no_grad():
            for m in list(plug.controlnet_down_blocks) + [plug.controlnet_mid_block, plug.cond_head[-1]]:
                m.weight.normal_(0, 0.02)
        tr = NativeTrainer(unet, None, plugins=[(plug, 1e-4)])
        kw["plugin_input"] = dict(cond=torch.rand(B, 3, 512, 512, device=dev))
    else:
        tr = NativeTrainer(unet, [dict(layers=PATS, rank=16 if sdxl else 8)])
        with torch.no_grad():
            for blk in tr.bucket.blocks:
                blk.layer.W_up.normal_(0, 0.02)
        tr.bucket.pack()
    hw, cd = (128, 2048) if sdxl else (64, 768)
    lat = torch.randn(B, 4, hw, hw, device=dev); ehs = torch.randn(B, 77, cd, device=dev).to(BF)
    if sdxl:
        kw["added_cond_kwargs"] = dict(text_embeds=torch.randn(B, 1280, device=dev), time_ids=torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B, device=dev))
    tr.train_one_step(lat, ehs, **kw)          # warm-up (lazy packing)
    K.lib().hcp_debug_gemm_table_stats(None, None)
    K.TRACE = []
    tr.train_one_step(lat, ehs, **kw)
    keys = collections.Counter(K.TRACE)
    K.TRACE = None
    import ctypes
    h, m = ctypes.c_long(), ctypes.c_long()
    K.lib().hcp_debug_gemm_table_stats(ctypes.byref(h), ctypes.byref(m))
    print(f"dispatch table during the traced step(s): {h.value} hits, {m.value} misses", flush=True)
    del tr, unet
    torch.cuda.empty_cache()
    return keys


def sweep(fn, nk1, cfgs=range(len(CFG_NAMES))):
    res = {}
    K.lib().hcp_debug_set_gemm_loaders(0)        # the MAIN table: no loader waves (tools/tune_loaders.py sweeps those on top of it)
    for cid in cfgs:
        for s in (1, 2, 4, 8, 16):
            if s > 1 and nk1 // s < 4:
                continue
            K.lib().hcp_debug_set_gemm_config(cid + 16 * s)
            try:
                res[f"{CFG_NAMES[cid]}/s{s}"] = round(timeit(fn), 1)
            except Exception:  # noqa: BLE001
                pass
    K.lib().hcp_debug_set_gemm_config(-1)
    heur = round(timeit(fn), 1)                  # what the main table / heuristic picks today (still without loaders)
    K.lib().hcp_debug_set_gemm_loaders(-1)
    return res, heur


def choose(res):
    best = min((v, k) for k, v in res.items())
    unsplit = min((v, k) for k, v in res.items() if k.endswith("/s1"))
    return unsplit if unsplit[0] <= 1.04 * best[0] else best      # same rule as gen_gemm_table.py


def main():
    workload = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else {"sdxl": 2, "dreambooth": 2}.get(workload, 4)
    keys = trace(workload, B)
    print(f"{workload} bs{B}: {len(keys)} distinct GEMM-family shapes, {sum(keys.values())} launches per step", flush=True)
    out = []
    saved = 0.0
    for key, cnt in sorted(keys.items(), key=lambda kv: -kv[1]):
        kind = key[0]
        if kind == "gemm":
            _, M, N, Kd, k2 = key
            if N % 4 or Kd % 8:
                continue
            a, b = rnd(M, Kd), rnd(N, Kd)
            a2, b2 = (rnd(M, 32), rnd(N, 32)) if k2 else (None, None)
            o = torch.empty(M, N, dtype=BF, device=dev)
            res, heur = sweep(lambda: K.gemm(a, b, a2=a2, b2=b2, out=o), Kd // 64)
            us, cfg = choose(res)
            tile, s = cfg.split("/s")
            out.append(dict(mode=0, M=M, N=N, K=Kd, has_k2=k2, stride=1, up=0, cfg=CFG_NAMES.index(tile), split=int(s), us=us, heur=heur, count=cnt))
        elif kind == "lora":
            _, M, N, Kd = key
            a, b, l, e = rnd(M, Kd), rnd(N, Kd), rnd(32, Kd), rnd(N, 32)

            def two():
                t = K.gemm(a, l)
                return K.gemm(a, b, a2=t, b2=e)
            K.lib().hcp_debug_set_gemm_loaders(0)
            heur = round(timeit(lambda: K.gemm_lora(a, b, l, e)), 1)
            t_two = round(timeit(two), 1)
            res = {}
            for cid in (0, 1, 2, 3, 4, 5, 6, 8, 9, 12, 13, 14, 15):
                K.lib().hcp_debug_set_gemm_config(cid + 16)
                res[cid] = round(timeit(lambda: K.gemm_lora(a, b, l, e)), 1)
            K.lib().hcp_debug_set_gemm_config(-1)
            K.lib().hcp_debug_set_gemm_loaders(-1)
            us, cid = min((v, k) for k, v in res.items())
            if t_two < us:
                us, cid = t_two, -1
            out.append(dict(mode=3, M=M, N=N, K=Kd, has_k2=1, stride=1, up=0, cfg=cid, split=1, us=us, heur=heur, count=cnt))
        else:
            _, mode, Bn, Hs, Ws, C1, C2, cout, stride, up, Ho, Wo, ext = key
            if ext:
                continue
            x1 = rnd(Bn, Hs, Ws, C1); x2 = rnd(Bn, Hs, Ws, C2) if C2 else None
            wp = rnd(cout, 3, 3, C1 + C2)
            fn = (lambda: K.conv3x3(x1, wp, cout, x2=x2, stride=stride, upsample=bool(up))) if mode == 0 else \
                 (lambda: K.conv3x3(x1, wp, cout, mode=1, stride=stride, out_hw=(Ho, Wo)))
            res, heur = sweep(fn, 9 * (C1 + C2) // 64)
            us, cfg = choose(res)
            tile, s = cfg.split("/s")
            out.append(dict(mode=1 if mode == 0 else 2, M=Bn * Ho * Wo, N=cout, K=9 * (C1 + C2), has_k2=0, stride=stride, up=up,
                            cfg=CFG_NAMES.index(tile), split=int(s), us=us, heur=heur, count=cnt))
        e = out[-1]
        saved += (e["heur"] - e["us"]) * cnt
        print(f"{key} x{cnt}: heuristic {e['heur']} us -> best {e['us']} us (cfg {e['cfg']} split {e['split']})", flush=True)
    print(f"potential saving per step: {saved / 1e3:.2f} ms", flush=True)
    root = os.environ.get("GRAFT_REPO_ROOT", ROOT)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", f"tune_extra_{workload}.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
