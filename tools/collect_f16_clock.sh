#!/usr/bin/env bash
# fp16 vs bf16 on the three MFMA kernels of tools/pmc_targets.py: the same instruction stream, busy cycles and active cycles
# per launch for both 16-bit types, next to the kernel durations of the same pass (gpurun -- 'bash tools/collect_f16_clock.sh').
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run this through gpurun}"
OUT=gpurun_out/f16clock
rm -rf "$OUT"; mkdir -p "$OUT"
for dt in bf16 f16; do
  export DT=$dt
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$dt" -- python3 tools/pmc_targets.py > "$OUT/$dt.log" 2>&1
done
python3 tools/f16_clock_table.py "$OUT" | tee "$OUT/table.txt"
