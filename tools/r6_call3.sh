#!/bin/bash
# round 6, GPU call 3: GEGLU forward in the projection epilogue — kernel / model tests, SDXL parity, step cost (SD1.5 headline + SDXL), step profile
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c3
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 1200 python -m pytest tests/test_kernels.py tests/test_model.py tests/test_abi.py -m gpu -q -x -k "geglu or hi_lo or tiny or abi or gemm or full_size" -rP -p no:cacheprovider > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log)
tail -3 $out/tests.log
timeout 1500 python tools/lab/sdxl_stream_ab.py > $out/sdxl_stream_ab.txt 2>&1
grep -E "====|sdxl b2|gates" $out/sdxl_stream_ab.txt
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $out/bench_sd15.json 2> $out/bench_sd15.err
python bench.py --workload sdxl --no-cpu-baseline --steps 30 --warmup 8 > $out/bench_sdxl.json 2> $out/bench_sdxl.err
python bench.py --workload sdxl --no-cpu-baseline --residual-stream off --steps 30 --warmup 8 > $out/bench_sdxl_stream_off.json 2> $out/bench_sdxl_stream_off.err
bash tools/step_profile.sh r6c3/step_sd15 > /dev/null 2>&1
rm -rf $out/step_sd15
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d.get("final_loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -16 $out/step_sd15_summary.md
