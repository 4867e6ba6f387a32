#!/bin/bash
# round 6, GPU call 10: the whole GPU suite + smoke on the final sources (after the DAPP container change), then the default bench line
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c10
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/gpu_tests.txt 2>&1; echo "rc=$?" >> $out/gpu_tests.txt)
tail -3 $out/gpu_tests.txt
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $out/gpu_tests.txt 2>&1; echo "smoke rc=$?" >> $out/gpu_tests.txt)
tail -2 $out/gpu_tests.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json | cut -c1-600
