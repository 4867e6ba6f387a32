#!/bin/bash
# round 6, GPU call 14: GELU from one hardware exponential (A&S 7.1.26) in the GEGLU kernels and epilogues — tests, headline twice, step profile, SDXL line
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c14
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 1500 python -m pytest tests/test_kernels.py tests/test_model.py tests/test_full_configs.py tests/test_trainer.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log)
tail -2 $out/tests.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-ckpt-line --steps 80 --warmup 20 > $out/bench_sd15_$i.json 2> $out/bench_sd15_$i.err; done
python bench.py --workload sdxl --no-cpu-baseline --steps 30 --warmup 8 > $out/bench_sdxl.json 2> /dev/null
bash tools/step_profile.sh r6c14/step_sd15 > /dev/null 2>&1
rm -rf $out/step_sd15
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c14/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d.get("final_loss"))
PY
head -12 $out/step_sd15_summary.md
grep "gn_" $out/step_sd15_summary.md
