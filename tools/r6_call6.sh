#!/bin/bash
# round 6, GPU call 6: tile-epilogue rule in the step (product rule vs nowhere vs everywhere), full GPU suite, step profile
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c6
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
for mode in -1 0 1 -1 0; do
  python tools/lab/epilogue_ab.py step $mode --no-cpu-baseline --no-ckpt-line --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd15 epilogue mode $mode', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
for mode in -1 0 1; do
  python tools/lab/epilogue_ab.py step $mode --workload sdxl --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl epilogue mode $mode', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
(timeout 1700 python -X faulthandler -m pytest tests -m gpu -v -rP -p no:cacheprovider > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log)
tail -3 $out/gpu_tests.log
grep -E "sdxl b2|FAILED|ERROR" $out/gpu_tests.log | head
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_sd15.json 2> $out/bench_sd15.err
bash tools/step_profile.sh r6c6/step_sd15 > /dev/null 2>&1
rm -rf $out/step_sd15
head -16 $out/step_sd15_summary.md
python -c "import json; d=json.loads(open('$out/bench_sd15.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])"
