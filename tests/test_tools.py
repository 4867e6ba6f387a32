"""The measurement tooling behind profiles/ (SURVEY §8 d): tools/prof_step_summary.py must separate the problems that share one
(kernel template, launch geometry) row — a kernel trace carries no arguments, but every step replays the same launch sequence — and
tools/pmc_roofline.py must pick the roofline kernels' own clusters out of such a row (round 4: the conv row at 256 workgroups averaged
the C320, C640 and C960 convolutions into a '47.8 us in-step' figure that described none of them)."""
import csv
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_summary_clusters_launch_positions(tmp_path):
    conv = "void hcp_gemm::(anonymous namespace)::gemm_pp_kernel<128, 160, 1, false, 4>(hcp_gemm::GemmParams)"
    dq = "void hcp_attn::attn2_bwd_dq_kernel<40, 2, false, 515>(hcp_attn::AttnParams)"
    seq = [(conv, 256 * 768, 1, 768, d) for d in (37, 38, 61, 37, 90, 37, 61, 37, 37, 37)]
    seq += [(dq, 1024 * 256, 1, 256, d) for d in (190, 12, 191, 12)]
    seq += [("void filler_kernel(int)", 64, 1, 64, 5)] * 60 + [("void adamw_kernel(int)", 64 * 256, 1, 256, 5)]
    rnd = random.Random(0)
    rows, t = [], 0
    for _ in range(25):
        for n, gx, gy, wg, d in seq:
            dur = int(d * 1000 * (1 + rnd.uniform(-0.02, 0.02)))
            rows.append({"Kernel_Name": n, "Start_Timestamp": t, "End_Timestamp": t + dur, "Grid_Size_X": gx, "Grid_Size_Y": gy, "Workgroup_Size_X": wg})
            t += dur + 1500
    d = tmp_path / "trace" / "x"
    d.mkdir(parents=True)
    with open(d / "kt_kernel_trace.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    out = tmp_path / "summary.md"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_step_summary.py"), str(tmp_path / "trace"), str(out), "20"], check=True, capture_output=True)
    text = out.read_text()
    row = next(l for l in text.splitlines() if l.startswith("| hcp_gemm::gemm_pp_kernel<128, 160, 1, false, 4> | 256 | 1 |"))
    cols = [c.strip() for c in row.strip("|").split("|")]
    assert cols[4] == "10.0" and abs(float(cols[5]) - 47.2) < 1.0                 # the row's plain average mixes three problems ...
    cl = [c.split(" x") for c in cols[7].split(" . ")]
    assert [int(n) for _, n in cl] == [7, 2, 1] and abs(float(cl[0][0]) - 37.3) < 1.0 and abs(float(cl[2][0]) - 90.0) < 2.0   # ... the clusters do not
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_roofline
    rec = pmc_roofline.in_step_averages(str(out))
    assert abs(rec["conv3x3_c320_64x64_b4"] - 37.3) < 1.0            # fastest cluster of the conv row = the K = 2880 convolution
    assert abs(rec["attn_dq_b4_h8_n4096_d40"] - 190.5) < 3.0         # slowest cluster of the attention row = self-attention at 64x64


def test_committed_step_summary_carries_the_clusters():
    """profiles/r6_step_kernel_summary.md (what bench.py's avg_launch_us_in_step and roofline_family are read from) has the position
    column and the family table, and the committed roofline record agrees with it."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_roofline
    rec = pmc_roofline.in_step_averages(os.path.join(ROOT, "profiles", "r6_step_kernel_summary.md"))
    js = json.load(open(os.path.join(ROOT, "profiles", "pmc_roofline.json")))
    assert set(rec) >= {"conv3x3_c320_64x64_b4", "attn_fwd_b4_h8_n4096_d40", "attn_dq_b4_h8_n4096_d40", "attn_dkv_b4_h8_n4096_d40"}
    for k, v in rec.items():
        assert abs(js[k]["avg_launch_us_in_step"] - v) < 1e-6
    assert 30.0 < rec["conv3x3_c320_64x64_b4"] < 45.0
    fam = pmc_roofline.family_times(os.path.join(ROOT, "profiles", "r6_step_kernel_summary.md"))
    assert abs(js["families"]["gemm_ms"] - fam["gemm_ms"]) < 1e-6 and abs(js["families"]["attention_ms"] - fam["attention_ms"]) < 1e-6
    assert 8.0 < fam["gemm_ms"] < 14.0 and 3.0 < fam["attention_ms"] < 5.0 and 700 < fam["dispatches_per_step"] < 1000
    sys.path.insert(0, ROOT)
    import bench
    rf = bench.family_roofline(4)
    assert rf and 0.1 < rf["gemm_and_conv"]["frac"] < 0.4 and 0.1 < rf["attention"]["frac"] < 0.4
