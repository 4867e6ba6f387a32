#!/bin/bash
# GPU box: rocprofv3 kernel trace of the headline bench -> gpurun_out/<tag>_summary.md (per-kernel table) + per-family totals.
# usage: tools/step_profile.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/$tag -o kt -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ckpt-line "$@" > $root/gpurun_out/$tag.log 2>&1
cd $root
python tools/prof_step_summary.py gpurun_out/$tag gpurun_out/${tag}_summary.md 20
