"""GPU lab: BASELINE configs[3] (SDXL LoRA r16, B=2, 1024 px) against the fp32 oracle fixture with the transformer blocks' residual stream as
plain bf16 and as the (hi | lo) pair (unet.set_residual_stream) — the parity lines of tests/test_full_configs.py for both, same process.
   python tools/lab/sdxl_stream_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_full_configs as T

orig = T._native_full
for mode in (False, True):
    def patched(dev, cfg=None, _m=mode):
        nat = orig(dev, cfg) if cfg is not None else orig(dev)
        nat.set_residual_stream(_m)
        return nat
    T._native_full = patched
    print(f"==== residual stream {'(hi | lo) pair' if mode else 'bf16'}", flush=True)
    try:
        T.test_sdxl_full_size_b2_1024px_full_lora_gradient_vs_golden()
        print("gates: passed", flush=True)
    except AssertionError as e:
        print(f"gates: FAILED {str(e)[:300]}", flush=True)
