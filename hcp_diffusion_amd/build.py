"""Build libhcp_mi355x.so (gfx950 only) from csrc/*.hip with hipcc, in-tree.

No CPU fallback exists: if hipcc is missing this raises.  `python -m hcp_diffusion_amd.build`
or `__graft_entry__.build()` call `build_product()`.
"""
import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(__file__).resolve().parent / "libhcp_mi355x.so"
TOOLS_LIB_PATH = Path(__file__).resolve().parent / "libhcp_mi355x_tools.so"   # same kernels + the hcp_debug_* tuning hooks (-DHCP_TOOLS)
SOURCES = ["runtime.hip", "gemm.hip", "gemm_pp.hip", "conv_patch.hip", "attention.hip", "norm.hip", "pointwise.hip", "lora.hip", "optim.hip", "wgrad.hip", "pack.hip", "comm.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libhcp_mi355x.so cannot be built (there is no CPU fallback)")


def _stale(out: Path, deps):
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_product(force: bool = False, verbose: bool = False, tools: bool = False) -> Path:
    """tools=True: the tuning build (libhcp_mi355x_tools.so) that tools/*.py and the variant-coverage tests load explicitly;
    the package itself only ever opens the product library."""
    hipcc = _hipcc()
    headers = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc"))
    objs = []
    bdir = CSRC / "build" / ("tools" if tools else ".")
    bdir.mkdir(parents=True, exist_ok=True)
    lib_path = TOOLS_LIB_PATH if tools else LIB_PATH
    procs = []
    for s in SOURCES:
        src = CSRC / s
        if not src.exists():
            raise RuntimeError(f"missing kernel source {src}")
        obj = bdir / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, *headers]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + (["-DHCP_TOOLS"] if tools else []) + \
                  ["-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    if force or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib_path)] + [str(o) for o in objs] + ["-ldl"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return lib_path


def build_tools(force: bool = False, verbose: bool = False) -> Path:
    return build_product(force, verbose, tools=True)


if __name__ == "__main__":
    print(build_product(verbose=True))
    print(build_tools(verbose=True))
