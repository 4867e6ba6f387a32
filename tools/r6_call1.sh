#!/bin/bash
# round 6, first GPU call: box identity, GPU suite on the atomics-free code, headline bench, grouped-wgrad lab, step profile
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c1
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 1700 python -X faulthandler -m pytest tests -m gpu -v -rP -p no:cacheprovider > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log)
tail -3 $out/gpu_tests.log
python bench.py --steps 100 --warmup 20 > $out/bench_sd15.json 2> $out/bench_sd15.err
tail -c 600 $out/bench_sd15.json
python tools/lab/wgrad_grouped_bench.py sd15 > $out/wgrad_sd15.txt 2>&1
python tools/lab/wgrad_grouped_bench.py sdxl > $out/wgrad_sdxl.txt 2>&1
cat $out/wgrad_sd15.txt
bash tools/step_profile.sh r6c1/step_sd15 > /dev/null 2>&1
rm -rf $out/step_sd15
