#!/bin/bash
# GPU box, round 5 call 2 (as run, the split was then the default and the switch was HCP_LAB_NO_T_SPLIT=1; the switch is HCP_T_SPLIT=1 now): full GPU suite (with the printed parity lines) + same-box A/B benches of the split T and the chunked wgrad.
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r5b; mkdir -p $out; cd $root
(timeout 900 python -X faulthandler -m pytest tests -m gpu -q -rP -p no:cacheprovider > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log)
grep -E "^\[|passed|failed|rc=" $out/gpu_tests.log | tail -40
b() { tag=$1; shift; python bench.py --no-cpu-baseline "$@" > $out/bench_$tag.json 2> $out/bench_$tag.err; python - $out/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d.get("ms_per_step"), "ms", {k: (d[k].get("ms_per_step") if isinstance(d.get(k), dict) else None) for k in ("grad_ckpt_on", "frozen_te_in_step", "seam_graph")}, (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
b sd15
HCP_T_SPLIT=1 b sd15_split --no-ckpt-line
b sd15_chunk32 --no-ckpt-line --wgrad-chunk 32
b sd15_chunk64 --no-ckpt-line --wgrad-chunk 64
b sd15_b --no-ckpt-line
b sdxl --workload sdxl --steps 30 --warmup 8
HCP_T_SPLIT=1 b sdxl_split --workload sdxl --steps 30 --warmup 8
b sdxl_chunk --workload sdxl --steps 30 --warmup 8 --wgrad-chunk 100
