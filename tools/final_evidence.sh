#!/bin/bash
# GPU box: the round's evidence in one call -> gpurun_out/<tag>/ (copied into profiles/ afterwards).
# usage: tools/final_evidence.sh <tag>
tag=${1:-final}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>/dev/null      # which box produced this evidence (unique id, partitions, driver, firmware)
(timeout 1500 python -X faulthandler -m pytest tests -m gpu -v -rP -p no:cacheprovider > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log)   # verbose + faulthandler: a crash names its test; -rP: the parity lines the full-size tests print
tail -2 $out/gpu_tests.log
python bench.py --steps 100 --warmup 20 > $out/bench_sd15.json 2> $out/bench_sd15.err
for w in sdxl dreambooth controlnet sd15te; do
  python bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err
done
python bench.py --seam > $out/bench_seam_sd15.json 2>/dev/null
python bench.py --seam --seam-graph > $out/bench_seam_graph_sd15.json 2>/dev/null
python bench.py --seam --workload dreambooth > $out/bench_seam_dreambooth.json 2>/dev/null
python bench.py --seam --seam-graph --workload dreambooth > $out/bench_seam_graph_dreambooth.json 2>/dev/null
bash tools/step_profile.sh $tag/step_sd15 > /dev/null 2>&1
bash tools/step_profile.sh $tag/step_sdxl --workload sdxl > /dev/null 2>&1
rm -rf $out/step_sd15 $out/step_sdxl          # the raw traces (tens of MB); the summaries stay
# counter passes of the roofline kernels (one counter group per run, never combined with tracing domains) + the record bench.py reads
bash tools/pmc_passes.sh $out/pmc -- python tools/pmc_roofline_target.py > /dev/null 2>&1
python tools/pmc_summary.py $out/pmc > $out/pmc_summary.md 2>/dev/null
python tools/pmc_roofline.py $out/pmc $out/pmc_roofline.json $out/step_sd15_summary.md > /dev/null 2>&1
rm -rf $out/pmc/p*/*/*.db 2>/dev/null
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("avg_launch_us"), (d.get("roofline_attention") or {}).get("avg_launch_us"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
if grep -l '"final_loss": NaN' $out/bench_*.json $out/*.log 2>/dev/null; then echo "!!! NaN final_loss in the files above"; else echo "no NaN final_loss in any bench line"; fi
