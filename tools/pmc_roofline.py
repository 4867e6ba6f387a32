"""tools/pmc_roofline.py PASSDIR [out.json]: per-launch HBM bytes and MFMA-busy of bench.py's roofline kernels from the
rocprofv3 counter passes of tools/pmc_passes.sh (target tools/pmc_roofline_target.py) -> profiles/pmc_roofline.json, which
bench.py reads for `roofline.traffic` (a profiler cannot run inside the benchmark process).
HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950: the counter tallies 128-byte requests as 64, MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024."""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict

KEYS = {"conv3x3_c320_64x64_b4": ("gemm_v2_kernel", "1, false"),      # MODE = 1 (forward conv), plain (no LoRA)
        "attn_fwd_b4_h8_n4096_d40": ("attn2_fwd_kernel<40", ""),
        "attn_dq_b4_h8_n4096_d40": ("attn2_bwd_dq_kernel<40", ""),
        "attn_dkv_b4_h8_n4096_d40": ("attn2_bwd_dkv_kernel<40", "")}


def main(passdir, out):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{passdir}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, (a, b) in KEYS.items():
                if a in r["Kernel_Name"] and b in r["Kernel_Name"]:
                    vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    vals[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    avg = lambda x: sum(x) / len(x)
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    rec = {}
    for key, v in vals.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        fetch, write = avg(v["FETCH_SIZE"]) * 1024 * 2, avg(v["WRITE_SIZE"]) * 1024
        e = {"hbm_bytes_per_launch": round(fetch + write), "hbm_fetch_bytes": round(fetch), "hbm_write_bytes": round(write),
             "us_under_pmc": round(avg(v["_us"]), 1), "git": git, "source": f"rocprofv3 --pmc passes, {passdir}"}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            e["mfma_busy"] = round(avg(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (avg(v["GRBM_GUI_ACTIVE"]) / 8 * 1024), 3)
        rec[key] = e
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_roofline.json")
