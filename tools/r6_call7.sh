#!/bin/bash
# round 6, GPU call 7: conv_patch.hip — correctness on the GPU, per-shape A/B, whole-step A/B
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c7
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv" -p no:cacheprovider > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log)
tail -3 $out/tests.log
timeout 900 python tools/lab/conv_patch_ab.py > $out/conv_patch_ab.txt 2>&1
grep -v amdgpu.ids $out/conv_patch_ab.txt
for on in 1 0 1 0; do
  python tools/lab/conv_patch_ab.py step $on --no-cpu-baseline --no-ckpt-line --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd15 conv patch $on', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
for on in 1 0; do
  python tools/lab/conv_patch_ab.py step $on --workload sdxl --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl conv patch $on', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
