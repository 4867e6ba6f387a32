cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_fix
python tools/lab/attn_ab.py tools/probes/libhcp_attn_old.so > gpurun_out/attn_fix/ab.txt 2>&1
cat gpurun_out/attn_fix/ab.txt
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "attention or attn" -x 2>&1 | tail -3 | tee gpurun_out/attn_fix/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/attn_fix/prof -o t -- python $GRAFT_REPO_ROOT/tools/attn_pmc_target.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/attn_fix/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn' in r['Name'] or 'gemm' in r['Name']:
            print(r['Name'][:90], r['Calls'], r['AverageNs'], r.get('MinNs'))
PY
rm -rf gpurun_out/attn_fix/prof/*/*.db
