"""Fill (almost) all of HBM with 0xFF bytes (NaN as fp32 / bf16) and exit: the NEXT process's fresh hipMalloc segments — including the private
pools of captured hipGraphs, which torch's caching allocator never recycles from the default pool — then start out as NaN unless the
driver scrubs freed VRAM.  Used before tools/diag/nan_hunt.py / bench.py to expose reads of memory nobody wrote."""
import torch
free, total = torch.cuda.mem_get_info()
n = int(free * 0.92) // (4 * 2 ** 28)
xs = [torch.full((2 ** 28,), -1, dtype=torch.int32, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
print(f"poisoned {n * 2 ** 30 / 2 ** 30:.0f} GiB of {total / 2 ** 30:.0f} GiB", flush=True)
