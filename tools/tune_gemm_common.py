import torch

BF = torch.bfloat16
dev = torch.device("cuda:0")
CFG_NAMES = ["128x128", "128x64", "64x64", "128x160", "64x160", "256x128", "256x160", "128x320", "128x160w8s3", "128x160w4s3", "256x160w8s3", "128x320w16", "256x160w16", "128x160w8", "64x160w8", "128x128w8"]


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
