"""bench.py's contract with the driver: the launcher guard (one rank per GPU or nothing) on the CPU; on the GPU one short run whose
single JSON line carries every field the contract names, the roofline record of the dominant kernel and the workload name."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_bench_refuses_a_world_that_is_not_one_rank_per_gpu():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.slow
@pytest.mark.parametrize("n", [2, 4])
def test_bench_launcher_runs_n_ranks_end_to_end(n):
    """`python bench.py --gpus N` is its own launcher (re-executes under torch.distributed.run, one rank per device).  Through the test
    launcher tests/emu/bench_emu.py (bench.main(emu=True) on the interpreter build) the very same code path — exec, rendezvous on 127.0.0.1, process group, NativeTrainer with the exchange, timed
    loop, MAX over ranks, ONE JSON line from rank 0 — runs on the CPU interpreter with gloo: the first 2-rank execution of this file
    must not be the driver's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "emu" / "bench_emu.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "1", "--rank-lora", "4"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["rccl_ranks"] == n and d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == n
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0 and "emu" in d["data"]
    assert d["loss_finite"] is True and d["final_loss"] > 0 and "invalid" not in d        # (round 5: the line says itself whether its loss was finite)


@pytest.mark.slow
def test_bench_launcher_two_ranks_sharded_overlapped_bf16_exchange():
    """The host-bucket path of the same launcher: `--workload dreambooth --exchange overlap-bf16` on 2 gloo ranks of the interpreter —
    sharded optimizer, chunks reduce-scattered from backward hooks, bf16 gradient and parameter wires — end to end through bench.py."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "emu" / "bench_emu.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", "--workload", "dreambooth",
                        "--exchange", "overlap-bf16"], env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["exchange"] == "overlap-bf16" and d["final_loss"] > 0


def test_abi_comm_bootstrap_falls_back_loudly_when_rccl_cannot_start(monkeypatch):
    """AbiComm.from_torch_store on a machine where the C-ABI communicator cannot be created (here: the interpreter build refuses
    world > 1... emulated with a world of 1 and a failing init) warns and hands back torch.distributed collectives instead of raising
    on one rank while the others wait."""
    import torch
    import torch.distributed as dist
    from hcp_diffusion_amd import comm as C
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 2000))
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        monkeypatch.setattr(C.AbiComm, "new_unique_id", staticmethod(lambda: b"\0" * 128))

        def boom(self, *a, **k):
            raise RuntimeError("hcp_comm_init: librccl.so not found")
        monkeypatch.setattr(C.AbiComm, "__init__", boom)
        with pytest.warns(RuntimeWarning, match="falling back to torch.distributed"):
            c = C.AbiComm.from_torch_store(torch.device("cpu"))
        assert isinstance(c, C.TorchComm) and c.world == 1
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-ckpt-line"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "loss_finite"):
        assert k in d, k
    assert d["loss_finite"] is True and "invalid" not in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "images/sec" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "SD1.5" in d["config"]["workload"] and d["config"]["global_batch"] == 4
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]          # images/s is the whole job over the timed region
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > 1e6
