#!/bin/bash
# round 6, GPU call 2: the (hi | lo) residual stream — kernel tests, SDXL parity A/B, SDXL / SD1.5 step cost
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r6c2
mkdir -p $out
cd $root
python tools/box_info.py > $out/box.json 2>&1
(timeout 900 python -m pytest tests/test_kernels.py tests/test_model.py tests/test_abi.py -m gpu -q -x -k "hi_lo or layernorm or merged_lora or tiny_sdxl or abi or gemm" -rP -p no:cacheprovider > $out/tests_stream.log 2>&1; echo "rc=$?" >> $out/tests_stream.log)
tail -3 $out/tests_stream.log
timeout 1500 python tools/lab/sdxl_stream_ab.py > $out/sdxl_stream_ab.txt 2>&1
grep -E "====|sdxl b2|gates" $out/sdxl_stream_ab.txt
for m in off on; do
  python bench.py --workload sdxl --no-cpu-baseline --residual-stream $m --steps 30 --warmup 8 > $out/bench_sdxl_stream_$m.json 2> $out/bench_sdxl_stream_$m.err
done
for m in off on; do
  python bench.py --no-cpu-baseline --no-ckpt-line --residual-stream $m --steps 60 --warmup 15 > $out/bench_sd15_stream_$m.json 2> $out/bench_sd15_stream_$m.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c2/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d.get("final_loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
