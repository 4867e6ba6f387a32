#!/bin/bash
# CPU-only: the kernel sources compiled for the wave64 interpreter (tests/emu) WITH AddressSanitizer, and the CPU test-suite run on it.
# torch's CPU tensors are malloc'd, so ASan red-zones every operand: a kernel that reads or writes one byte past a tensor stops with the
# source line (the plain interpreter, like the GPU most of the time, lets small overruns pass silently; on the GPU they fault only when
# the allocation ends a mapped region).  Buffer-addressed loads / LDS-DMA go through the interpreter's bounds-checked helpers, exactly
# as the hardware range check does, so only the kernels' raw pointer accesses are under test.
#   usage: tools/diag/emu_asan.sh [pytest args...]        (default: the kernel / model / trainer files, -n 8)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${HCP_EMU_ASAN_DIR:-/tmp/emu_asan}
mkdir -p "$out" /tmp/asan_logs
cxx=/opt/rocm/lib/llvm/bin/clang++
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
objs=""
# (comm.hip binds RCCL and is the one product source the interpreter does not compile: tests/emu/hcp_emu_comm.cpp stands in, as in build_emu.py)
for s in $(python -c "import sys; sys.path.insert(0, '$root'); from hcp_diffusion_amd.build import SOURCES; print(' '.join(x for x in SOURCES if x != 'comm.hip'))"); do
  o=$out/${s%.hip}.o; objs="$objs $o"
  if [ ! -f "$o" ] || [ -n "$(find "$root/hcp_diffusion_amd/csrc" "$root/tests/emu" -newer "$o" \( -name '*.h' -o -name '*.inc' -o -name "$s" \) | head -1)" ]; then
    $cxx -x c++ -std=c++17 -O1 -g -fPIC -DHCP_EMU -DHCP_TOOLS -ffp-contract=off -fsanitize=address -fno-omit-frame-pointer -shared-libasan \
         -fvisibility=hidden -Wno-unused-function -Wno-unknown-attributes -I"$root/tests/emu" -I"$root/hcp_diffusion_amd/csrc" \
         -c "$root/hcp_diffusion_amd/csrc/$s" -o "$o" &
  fi
done
$cxx -x c++ -std=c++17 -O1 -g -fPIC -DHCP_EMU -DHCP_TOOLS -fsanitize=address -fno-omit-frame-pointer -shared-libasan -fvisibility=hidden \
     -I"$root/tests/emu" -I"$root/hcp_diffusion_amd/csrc" -c "$root/tests/emu/hcp_emu.cpp" -o "$out/hcp_emu.o" &
$cxx -x c++ -std=c++17 -O1 -g -fPIC -DHCP_EMU -DHCP_TOOLS -fsanitize=address -fno-omit-frame-pointer -shared-libasan -fvisibility=hidden \
     -I"$root/tests/emu" -I"$root/hcp_diffusion_amd/csrc" -c "$root/tests/emu/hcp_emu_comm.cpp" -o "$out/hcp_emu_comm.o" &
wait
$cxx -shared -fPIC -fsanitize=address -shared-libasan -o "$out/libhcp_emu_asan.so" $objs "$out/hcp_emu.o" "$out/hcp_emu_comm.o"
cd "$root"
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(tests/test_kernels.py tests/test_model.py tests/test_vae.py tests/test_text_encoder.py tests/test_trainer.py tests/test_sampler.py tests/test_graphed.py tests/test_ckpt.py -n 8)
HCP_EMU_LIB="$out/libhcp_emu_asan.so" LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:log_path=/tmp/asan_logs/asan \
  python -m pytest -q -m "not gpu" "${args[@]}"
echo "ASan reports (if any): /tmp/asan_logs/"
