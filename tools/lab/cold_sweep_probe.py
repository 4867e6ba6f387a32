"""GPU lab (tools library): do the warm-loop winners of the GEMM dispatch table stay winners under in-step conditions (operands not in L2:
weights from HBM, activations from the Infinity Cache)?  Each timed launch is preceded by a 512 MB fill (evicts L2 + MALL) and a copy that
re-writes the activation operand (so it sits where a producer kernel would have left it); the fill + copy time is measured alone and
subtracted."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd import _lib
K._set_backend_for_tests(_lib.load_tools())
dev = torch.device("cuda:0")
CFG = ["128x128", "128x64", "64x64", "128x160", "64x160", "256x128", "256x160", "128x320", "128x160w8s3", "128x160w4s3", "256x160w8s3", "128x320w16", "256x160w16", "128x160w8", "64x160w8", "128x128w8"]
def rnd(*s): return (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
trash = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def gtime(body, n=10):
    for _ in range(2): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3
def cold_warm(fn, act, act_src):
    def pre(): trash.zero_(); act.copy_(act_src)
    base = gtime(pre)
    cold = gtime(lambda: (pre(), fn())) - base
    warm = gtime(fn, n=20)
    return cold, warm
x = rnd(4, 64, 64, 320); xs = x.clone(); w = rnd(320, 3, 3, 320)
a1 = rnd(16384, 320); a1s = a1.clone(); b1, l1, e1 = rnd(320, 320), rnd(32, 320), rnd(320, 32)
a2 = rnd(1024, 1280); a2s = a2.clone(); b2, l2, e2 = rnd(1280, 1280), rnd(32, 1280), rnd(1280, 32)
a3 = rnd(4096, 2560); a3s = a3.clone(); b3, l3, e3 = rnd(640, 2560), rnd(32, 2560), rnd(640, 32)
cases = [("conv C320@64^2", lambda: K.conv3x3(x, w, 320), x, xs, [(13, 1), (14, 1), (3, 1), (13, 2), (12, 2)]),
         ("fused-LoRA M16384 N320 K320", lambda: K.gemm_lora(a1, b1, l1, e1), a1, a1s, [(13, 1), (14, 1), (15, 1), (4, 1), (1, 1)]),
         ("fused-LoRA M1024 N1280 K1280", lambda: K.gemm_lora(a2, b2, l2, e2), a2, a2s, [(13, 1), (14, 1), (15, 1), (4, 1), (2, 1)]),
         ("fused-LoRA M4096 N640 K2560", lambda: K.gemm_lora(a3, b3, l3, e3), a3, a3s, [(13, 1), (14, 1), (15, 1), (4, 1), (14, 2)])]
for name, fn, act, src, cfgs in cases:
    K.lib().hcp_debug_set_gemm_config(-1); K.lib().hcp_debug_set_gemm_loaders(-1)
    c, wv = cold_warm(fn, act, src)
    print(f"{name}: dispatched        cold {c:6.1f} warm {wv:6.1f}", flush=True)
    for cid, s in cfgs:
        for ld in ((1, 3, 4) if cid >= 13 else (0,)):
            K.lib().hcp_debug_set_gemm_config(cid + 16 * s); K.lib().hcp_debug_set_gemm_loaders(ld if cid >= 13 else -1)
            try:
                c, wv = cold_warm(fn, act, src)
                print(f"   {CFG[cid]:12s}/s{s} ring {2 if ld == 1 else ld if ld else '-'}: cold {c:6.1f} warm {wv:6.1f}", flush=True)
            except Exception as ex:  # noqa: BLE001
                print(f"   {CFG[cid]}/s{s} ld{ld}: failed {str(ex)[:60]}", flush=True)
K.lib().hcp_debug_set_gemm_config(-1); K.lib().hcp_debug_set_gemm_loaders(-1)
