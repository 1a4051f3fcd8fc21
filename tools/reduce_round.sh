#!/usr/bin/env bash
# Reduce the raw outputs of tools/collect_round.sh (gpurun_out/<tag>{p,t,x}/) into the tracked files profiles/<tag>_* (CPU, build container).
set -euo pipefail
TAG="${1:-r05}"
cd "$(dirname "$0")/.."
export ROUND_TAG="$TAG"
P="gpurun_out/${TAG}p"; T="gpurun_out/${TAG}t"; X="gpurun_out/${TAG}x"
kt=$(ls $P/step/*/*kernel_trace.csv | head -1)
python3 tools/summarize_profile.py --trace "$kt" --out "profiles/${TAG}_step_kernel_trace_summary.csv"
python3 tools/summarize_profile.py --one-step "$kt" --out "profiles/${TAG}_one_step_breakdown.csv"
cp "$(ls $P/step/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_step_b32_kernel_stats.csv"
python3 tools/summarize_profile.py --derive $(ls $P/fetch/*/*counter_collection.csv $P/write/*/*counter_collection.csv $P/mfma/*/*counter_collection.csv $P/sq/*/*counter_collection.csv) \
  --out "profiles/${TAG}_mfma_pmc.json" --trunk-out "profiles/${TAG}_trunk_pmc.json"
python3 tools/summarize_profile.py --traffic "$(ls $T/fetch/*/*counter_collection.csv | head -1)" "$(ls $T/write/*/*counter_collection.csv | head -1)" --steps 2 --out "profiles/${TAG}_step_pmc_traffic.md"
cp "$X/bench_default.json" "profiles/${TAG}_bench_default.log"
for f in layer_table configs ab_halo ab_cls_skip ab_march ab_dfirst ab_dlast determinism_stress soak; do cp "$X/$f.log" "profiles/${TAG}_$f.log"; done
for f in bench_cfg3 bench_cfg5 bench_fp8 bench_fp16_storage; do cp "$X/$f.json" "profiles/${TAG}_$f.json"; done
cp "$X/bench_rccl_rehearsal.json" "profiles/${TAG}_bench_rccl_rehearsal.log"
ls -la profiles/${TAG}_*
