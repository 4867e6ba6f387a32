"""Per-STEP kernel table from a rocprofv3 --kernel-trace csv:  python tools/prof_step_summary.py <dir> <out.md> [steps] [anchor-regex]

The trace also holds weight initialisation, warm-up and graph capture; the table keeps only the LAST `steps` training steps (cut at
the optimizer kernel that ends each step) and divides by `steps`, so "calls" and "ms" are per step of the replayed hipGraph.

A kernel template at one launch geometry can still serve SEVERAL problems (the conv template at 256 workgroups runs C320, C640 and C960
inputs: K = 2880 / 5760 / 8640); the trace has no kernel arguments, but every step replays the same launch sequence, so the j-th launch of
a step is the same problem in every step: the last column averages each POSITION over the steps and groups positions within 8 % of each
other ("37.6 x7 . 61.2 x2 . 90.7 x1" = three problems behind one row)."""
import collections
import csv
import glob
import re
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    anchor = re.compile(sys.argv[4] if len(sys.argv) > 4 else r"adamw")
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*", "", n)[:100]
        wg = max(1, int(r.get("Workgroup_Size_X", 1) or 1))
        shape = (int(r.get("Grid_Size_X", 0) or 0) // wg, int(r.get("Grid_Size_Y", 1) or 1), wg)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, shape))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if anchor.search(r[2])]
    # several optimizer launches may end one step (one per bucket): a step boundary = the last anchor of a run
    bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 50]
    assert len(bounds) > steps, f"only {len(bounds)} steps in the trace"
    lo, hi = bounds[-steps - 1] + 1, bounds[-1] + 1
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / steps / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    by_shape = collections.defaultdict(lambda: [0, 0.0])      # GEMM family per (kernel template, workgroups x split-K slabs, threads)
    for s, e, n, shape in sel:
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
        if "gemm_" in n or "conv_patch" in n or "attn2_" in n:
            by_shape[(n, shape)][0] += 1; by_shape[(n, shape)][1] += (e - s) / 1e3
    # per launch POSITION inside the step (same problem in every step): average over the steps, then cluster per geometry row
    per_step = (hi - lo) // steps
    by_pos = collections.defaultdict(list)
    if per_step * steps == hi - lo and all(sel[j][2] == sel[j + per_step][2] for j in range(0, hi - lo - per_step, 37)):
        for j in range(per_step):
            n, shape = sel[j][2], sel[j][3]
            if "gemm_" in n or "conv_patch" in n or "attn2_" in n:
                by_pos[(n, shape)].append(sum((sel[j + k * per_step][1] - sel[j + k * per_step][0]) for k in range(steps)) / steps / 1e3)

    def clusters(v):
        out = []
        for x in sorted(v):
            if out and x <= out[-1][0] * 1.08:
                out[-1][1].append(x)
            else:
                out.append([x, [x]])
        return " . ".join(f"{sum(c) / len(c):.1f} x{len(c)}" for _, c in out)
    tot = sum(v[1] for v in agg.values())
    fam = collections.defaultdict(lambda: [0, 0.0])
    for n, (c, t) in agg.items():
        k = ("attention" if "attn" in n else "gemm / implicit conv" if ("gemm_" in n or "conv_patch" in n) else "split-K reduce" if "splitk" in n else
             "GroupNorm" if n.startswith("gn_") else "LayerNorm" if n.startswith("ln_") else "GEGLU" if "geglu" in n else
             "weight gradients" if "wgrad" in n else "optimizer" if "adamw" in n else "torch (at::)" if "at::" in n else "other")
        fam[k][0] += c; fam[k][1] += t
    L = [f"last {steps} steps of the trace: {len(sel) / steps:.0f} dispatches per step, kernel time {tot / steps / 1e3:.3f} ms per step, "
         f"first start -> last end {wall:.3f} ms per step (profiler attached)", "",
         "| family | launches / step | ms / step | % |", "|---|---|---|---|"]
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        L.append(f"| {k} | {c / steps:.0f} | {t / steps / 1e3:.3f} | {100 * t / tot:.1f} |")
    L += ["", "| kernel | launches / step | ms / step | avg us | % |", "|---|---|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        L.append(f"| {n} | {c / steps:.1f} | {t / steps / 1e3:.3f} | {t / c:.1f} | {100 * t / tot:.1f} |")
    L += ["", "GEMM family and attention by launch geometry (one row = one problem shape class: tiles along M x N in `workgroups`, split-K slabs in `y`):", "",
          "| kernel | workgroups | y | threads | launches / step | avg us | ms / step | us by launch position (problems behind the row) |", "|---|---|---|---|---|---|---|---|"]
    for (n, (wgs, gy, thr)), (c, t) in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:90]:
        L.append(f"| {n} | {wgs} | {gy} | {thr} | {c / steps:.1f} | {t / c:.1f} | {t / steps / 1e3:.3f} | {clusters(by_pos.get((n, (wgs, gy, thr)), []))} |")
    open(out, "w").write("\n".join(L) + "\n")
    print("\n".join(L[:16]))


if __name__ == "__main__":
    main()
