"""TEST LAUNCHER (tests/test_bench.py): bench.py's own main() — launcher re-exec, rendezvous, process group, NativeTrainer with the exchange,
timed loop, MAX over ranks, ONE JSON line — on the CPU interpreter build of the kernels with gloo.  bench.py itself holds no test switch:
this file points hcp_diffusion_amd.kernels at the interpreter and calls bench.main(emu=True); when bench.py re-executes "itself" under
torch.distributed.run it re-executes sys.argv[0], i.e. this launcher."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    from conftest import emu_cdll
    from hcp_diffusion_amd import kernels as K
    K._set_backend_for_tests(emu_cdll())
    import bench
    bench.main(emu=True)
