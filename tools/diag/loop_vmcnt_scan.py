"""Scan a hipcc -S listing for COMPILER-inserted `s_waitcnt vmcnt(N)` inside loops of kernels that also issue LDS-DMA from inline asm.

Why: the LDS-DMA tile fills are hidden from the compiler's waitcnt pass (hcp_dma16 is inline asm), but the hardware counter is shared —
a compiler-inserted vmcnt wait for some VGPR-destination load, placed behind a DMA issue inside the main loop, also drains the tile
prefetch and the wave sits out the whole L2 -> LDS latency every iteration (found in the dK/dV attention kernel, round 4).

usage:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I hcp_diffusion_amd/csrc <file.hip> -o /tmp/x.s
        python tools/diag/loop_vmcnt_scan.py /tmp/x.s
Prints, per kernel, every compiler vmcnt wait that FOLLOWS an asm DMA issue of the same loop iteration (straight-line order within the
loop body) and still has MFMA work behind it before the kernel's own counted wait — a wait directly in front of the kernel's own
`s_waitcnt vmcnt` (statistics stored at the tile end) costs nothing and is not reported."""
import re
import subprocess
import sys


def demangle(s):
    try:
        return subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
    except Exception:
        return s


def scan(path):
    lines = open(path).read().split("\n")
    i, n, found = 0, len(lines), 0
    while i < n:
        m = re.match(r"^(_Z\w+):\s+; @", lines[i])
        if not m:
            i += 1
            continue
        name, j = m.group(1), i + 1
        in_asm, in_loop, dma_seen, hits = False, False, False, []
        while j < n and not lines[j].strip().startswith(".amdhsa_kernel") and not lines[j].startswith(".Lfunc_end"):
            t = lines[j].strip()
            if t.startswith(".LBB") or t.startswith("; %bb."):
                in_loop = "Loop" in t
                if "Loop Header" in t:
                    dma_seen = False                      # a new iteration starts here
            elif "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            elif in_loop:
                if in_asm and "buffer_load" in t and " lds" in t:
                    dma_seen = True
                elif in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:
                    dma_seen = False                      # the kernel's own counted wait: what is in flight after it is by design
                    hits = [h for h in hits if h[2] is None or h[2] > 0]
                    hits = [(a, b, None) if c is not None else (a, b, c) for a, b, c in hits]
                elif not in_asm and t.startswith("s_waitcnt") and "vmcnt" in t and dma_seen:
                    hits.append((j + 1, t, 0))
                elif "v_mfma" in t:
                    hits = [(a, b, c + 1) if c is not None else (a, b, c) for a, b, c in hits]
            j += 1
        hits = [h for h in hits if h[2] is None or h[2] > 0]
        if hits:
            found += 1
            print(demangle(name).replace("(hcp_attn::AttnParams)", "").replace("void ", ""))
            for ln, t, _ in hits[:6]:
                print("    line %d: %s" % (ln, t))
        i = j
    print("%d kernel(s) with a compiler vmcnt wait behind an in-loop DMA issue" % found)
    return found


if __name__ == "__main__":
    sys.exit(1 if scan(sys.argv[1]) else 0)
