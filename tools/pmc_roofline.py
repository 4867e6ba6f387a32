"""tools/pmc_roofline.py PASSDIR [out.json [step_summary.md]]: per-launch HBM bytes and MFMA-busy of bench.py's roofline kernels from the
rocprofv3 counter passes of tools/pmc_passes.sh (target tools/pmc_roofline_target.py) -> profiles/pmc_roofline.json, which
bench.py reads for `roofline.traffic` (a profiler cannot run inside the benchmark process).
HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950: the counter tallies 128-byte requests as 64, MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024."""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict

KEYS = {"conv3x3_c320_64x64_b4": ("gemm_pp_kernel<128, 160, 1, false", ""),      # MODE = 1 (forward conv), plain (no LoRA), ping-pong loop (conv_patch.hip takes the data gradients and split-K launches)
        "attn_fwd_b4_h8_n4096_d40": ("attn2_fwd_kernel<40", ""),
        "attn_dq_b4_h8_n4096_d40": ("attn2_bwd_dq_kernel<40", ""),
        "attn_dkv_b4_h8_n4096_d40": ("attn2_bwd_dkv_kernel<40", "")}


def in_step_averages(summary_md):
    """avg us of the roofline kernels INSIDE the captured step, from the per-geometry table tools/prof_step_summary.py writes: the conv
    template at 256 workgroups x 1 slab — its FASTEST position cluster (that row also holds the C640 / C960 -> 320 convolutions of the up
    blocks: K = 5760 / 8640; C320 -> 320 at 64x64, B 4 is the K = 2880 one) — and, for the attention templates, the geometry with the
    most time and its SLOWEST cluster (self-attention at 64x64; the cross-attention launches share the row)."""
    rows = []
    sec = False
    for line in open(summary_md):
        if "by launch geometry" in line:
            sec = True
            continue
        if sec and line.startswith("|") and not line.startswith("| kernel") and not line.startswith("|---"):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            cl = [float(x.split(" x")[0]) for x in c[7].split(" . ")] if len(c) > 7 and c[7] else []
            rows.append((c[0], int(c[1]), int(c[2]), float(c[5]), float(c[6]), cl))
    res = {}
    for key, (a, b) in KEYS.items():
        cand = [r for r in rows if a in r[0] and b in r[0] and (not key.startswith("conv") or (r[1] == 256 and r[2] == 1))]
        if cand:
            r = max(cand, key=lambda r: r[4])
            res[key] = (min(r[5]) if key.startswith("conv") else max(r[5])) if r[5] else r[3]
    return res


def family_times(summary_md):
    """ms per step of the kernel families and the launch count, from the head of tools/prof_step_summary.py's table (the committed step
    summary): what bench.py's `roofline_family` divides the families' algorithmic FLOPs by (VERDICT r5 next #4d: the `roofline` object
    shows the best kernel of its family, this one the family)."""
    import re
    fam, head = {}, open(summary_md).readline()
    for line in open(summary_md):
        m = re.match(r"\| ([^|]+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|$", line.strip())
        if m and m.group(1) not in ("family",):
            fam[m.group(1)] = (int(m.group(2)), float(m.group(3)))
        if line.startswith("| kernel"):
            break
    d = re.search(r"(\d+) dispatches per step, kernel time ([\d.]+) ms", head)
    return {"gemm_ms": fam.get("gemm / implicit conv", (0, 0.0))[1], "gemm_launches": fam.get("gemm / implicit conv", (0, 0.0))[0],
            "attention_ms": fam.get("attention", (0, 0.0))[1], "attention_launches": fam.get("attention", (0, 0.0))[0],
            "splitk_reduce_ms": fam.get("split-K reduce", (0, 0.0))[1],
            "dispatches_per_step": int(d.group(1)) if d else None, "kernel_ms_per_step": float(d.group(2)) if d else None, "source": summary_md}


def main(passdir, out, step_summary=None):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{passdir}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, (a, b) in KEYS.items():
                if a in r["Kernel_Name"] and b in r["Kernel_Name"]:
                    vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    vals[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    trace = defaultdict(list)                          # the kernel-trace pass of the same command (no counters attached)
    for f in glob.glob(f"{passdir}/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, (a, b) in KEYS.items():
                if a in r["Kernel_Name"] and b in r["Kernel_Name"]:
                    trace[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    avg = lambda x: sum(x) / len(x)
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    rec = {}
    for key, v in vals.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        fetch, write = avg(v["FETCH_SIZE"]) * 1024 * 2, avg(v["WRITE_SIZE"]) * 1024
        e = {"hbm_bytes_per_launch": round(fetch + write), "hbm_fetch_bytes": round(fetch), "hbm_write_bytes": round(write),
             "us_under_pmc": round(avg(v["_us"]), 1), "git": git, "source": f"rocprofv3 --pmc passes, {passdir}"}
        if trace.get(key):
            e["us_rocprof_trace"] = round(avg(trace[key]), 1)
            e["us_rocprof_trace_launches"] = [round(x, 1) for x in trace[key]]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            e["mfma_busy"] = round(avg(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (avg(v["GRBM_GUI_ACTIVE"]) / 8 * 1024), 3)
        rec[key] = e
    if step_summary:
        for key, us in in_step_averages(step_summary).items():
            if key in rec:
                rec[key]["avg_launch_us_in_step"] = us
                rec[key]["in_step_source"] = step_summary
    if step_summary:
        rec["families"] = dict(family_times(step_summary), git=git)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_roofline.json", sys.argv[3] if len(sys.argv) > 3 else None)
