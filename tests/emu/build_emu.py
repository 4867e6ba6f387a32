"""TEST-ONLY: compile csrc/*.hip as host C++ against tests/emu/hcp_emu.h (wave64 interpreter).

The result, tests/emu/libhcp_emu.so, exports the same C ABI as libhcp_mi355x.so but takes host
pointers.  It is loaded only by tests (tests/conftest.py), never by the package.
"""
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent.parent / "hcp_diffusion_amd" / "csrc"
LIB = HERE / "libhcp_emu.so"


def _cxx():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("host clang++ (ext_vector_type support) not found")


def build_emu(force=False):
    import sys
    sys.path.insert(0, str(HERE.parent.parent))
    from hcp_diffusion_amd.build import SOURCES
    # comm.hip (RCCL) is the one product source the interpreter does not compile: hcp_emu_comm.cpp stands in for a world of one rank
    srcs = [CSRC / s for s in SOURCES if s != "comm.hip"] + [HERE / "hcp_emu.cpp", HERE / "hcp_emu_comm.cpp"]
    deps = srcs + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [HERE / "hcp_emu.h"]
    if not force and LIB.exists() and all(d.stat().st_mtime <= LIB.stat().st_mtime for d in deps):
        return LIB
    bdir = HERE / "build"
    bdir.mkdir(exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        obj = bdir / (s.stem + ".o")
        objs.append(obj)
        if force or not obj.exists() or any(d.stat().st_mtime > obj.stat().st_mtime for d in [s] + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [HERE / "hcp_emu.h"]):
            cmd = [_cxx(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DHCP_EMU", "-DHCP_TOOLS", "-ffp-contract=off",
                   "-fvisibility=hidden", "-Wno-unused-function", "-Wno-unknown-attributes",
                   f"-I{HERE}", f"-I{CSRC}", "-c", str(s), "-o", str(obj)]
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"emu compile failed on {s}:\n{out.decode()}")
    r = subprocess.run([_cxx(), "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(r.stdout.decode())
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
