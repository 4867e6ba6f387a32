"""GPU diagnostic: the forced one-rank sharded + overlapped exchange against the plain trainer, repeated; on a mismatch list the
parameters that differ (flaky failure of tests/test_trainer.py::test_sharded_exchange_from_backward_on_one_rank[gpu-False])."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_trainer import _batch, _fix_noise, _native
from hcp_diffusion_amd.trainer import NativeTrainer
dev = torch.device("cuda:0")
data = [dict(**_batch(dev, 1)), dict(**_batch(dev, 2), loss_weight=0.5)]
def run(kw, steps=2):
    tr = NativeTrainer(_native(dev), None, lr=1e-3, train_cfg=[dict(layers=[""])], **kw)
    _fix_noise(tr, dev)
    for _ in range(steps):
        tr.train_data_list([dict(d) for d in data])
    torch.cuda.synchronize()
    return {n: p.detach().float().cpu().clone() for n, p in tr.unet.named_parameters()}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    a = run({})
    b = run(dict(shard_optimizer="force", overlap_exchange=True))
    c = run(dict(shard_optimizer="force"))
    for tag, other in (("overlap", b), ("sharded, no overlap", c)):
        bad = [(n, (a[n] - other[n]).abs().max().item(), a[n].numel()) for n in a if (a[n] - other[n]).abs().max().item() > 2e-4]
        print(f"rep {rep} {tag}: {len(bad)} tensors differ by > 2e-4" + ("" if not bad else ": " + "; ".join(f"{n} ({m:.1e}, {k} el)" for n, m, k in bad[:8])), flush=True)
