#!/bin/bash
# round 6, GPU call 15: smoke() and the default bench line on the final library with the final committed evidence record
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c15
mkdir -p $out
cd $root
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt)
tail -4 $out/smoke.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json | cut -c1-400
