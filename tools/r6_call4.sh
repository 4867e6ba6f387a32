#!/bin/bash
# round 6, GPU call 4: same-box A/B of the GEGLU product in the projection epilogue vs the stand-alone pass (SD1.5 headline, SDXL)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c4
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-ckpt-line --steps 80 --warmup 20 > $out/sd15_fused_$rep.json 2> $out/sd15_fused_$rep.err
  python tools/lab/bench_geglu_ab.py --no-cpu-baseline --no-ckpt-line --steps 80 --warmup 20 > $out/sd15_twopass_$rep.json 2> $out/sd15_twopass_$rep.err
done
python bench.py --workload sdxl --no-cpu-baseline --steps 30 --warmup 8 > $out/sdxl_fused.json 2> $out/sdxl_fused.err
python tools/lab/bench_geglu_ab.py --workload sdxl --no-cpu-baseline --steps 30 --warmup 8 > $out/sdxl_twopass.json 2> $out/sdxl_twopass.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c4/*.json")):
    if f.endswith("box.json"): continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d.get("final_loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
