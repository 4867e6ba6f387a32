#!/bin/bash
# tools/pmc_passes.sh OUTDIR -- CMD...   : rocprofv3 counter passes (one counter group per run, as MI355X_MICROARCH.md prescribes;
# never combined with sys/hip/hsa tracing) + a kernel-trace pass for durations.  Summarise with tools/pmc_summary.py OUTDIR.
set -u
out=$1; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES")
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/p$i" -o pmc -- "$@" > "$out/p$i.log" 2>&1 || echo "pass $i ($g) failed, see $out/p$i.log"
  i=$((i+1))
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o kt -- "$@" > "$out/trace.log" 2>&1 || echo "trace pass failed"
