#!/usr/bin/env bash
# rocprofv3 evidence of one round (run on the GPU box: gpurun -- 'bash tools/collect_profiles.sh r04 [trace|pmc|all]').
# Every pass writes into a FRESH directory under gpurun_out/<tag>p/ (a failed pass must not be summarised from old files);
# tools/summarize_profile.py turns the CSVs into the tracked files under profiles/.
# The program after `--` is always python3 itself (no env / bash -c hop: the profiler has initialised the GPU by then).
set -euo pipefail
TAG="${1:-r05}"
WHAT="${2:-all}"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}"
OUT="gpurun_out/${TAG}p"
mkdir -p "$OUT"
fresh() { rm -rf "$OUT/$1"; mkdir -p "$OUT/$1"; }
if [ "$WHAT" = "trace" ] || [ "$WHAT" = "all" ]; then
  fresh step
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mdct > "$OUT/step.log" 2>&1
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  fresh fetch; fresh write; fresh mfma; fresh sq
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/pmc_targets.py > "$OUT/fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/pmc_targets.py > "$OUT/write.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 tools/pmc_targets.py > "$OUT/mfma.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/sq" -- python3 tools/pmc_targets.py > "$OUT/sq.log" 2>&1
fi
find "$OUT" -name "*.csv" | head -30
tail -n 2 "$OUT"/*.log
