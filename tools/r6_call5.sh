#!/bin/bash
# round 6, GPU call 5: tile epilogue A/B (per shape + whole step), the host-stub reproducer (VERDICT r5 next #1c)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c5
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "tile_epilogue or geglu_fwd or hi_lo" -p no:cacheprovider > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log)
tail -2 $out/tests.log
timeout 900 python tools/lab/epilogue_ab.py > $out/epilogue_ab.txt 2>&1
cat $out/epilogue_ab.txt
for mode in 0 1 0 1; do
  python tools/lab/epilogue_ab.py step $mode --no-cpu-baseline --no-ckpt-line --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sd15 epilogue mode $mode', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
for mode in 0 1; do
  python tools/lab/epilogue_ab.py step $mode --workload sdxl --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sdxl epilogue mode $mode', d['ms_per_step'], d['final_loss'])" | tee -a $out/step_ab.txt
done
timeout 900 python tools/probes/stub_repro/run.py > $out/stub_repro.txt 2>&1
cat $out/stub_repro.txt | tail -5
