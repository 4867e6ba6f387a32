import torch

BF = torch.bfloat16
dev = torch.device("cuda:0")
CFG_NAMES = ["128x128", "128x64", "64x64", "128x160", "64x160", "256x128", "256x160", "128x320", "128x160w8s3", "128x160w4s3", "256x160w8s3", "128x320w16", "256x160w16", "128x160w8", "64x160w8", "128x128w8"]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def rnd(*s):
    return torch.randn(*s, device=dev).to(BF)
