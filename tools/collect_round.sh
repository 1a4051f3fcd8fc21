#!/usr/bin/env bash
# Everything profiles/<tag>_* holds, collected in ONE gpurun call on the build that is in the tree (round-4 review, weak 2b: a
# profile taken before a kernel fix is not a profile of the product):  gpurun -- 'bash tools/collect_round.sh r05'
# Raw outputs go to gpurun_out/<tag>{p,t,x}/ (scratch, merged back by gpurun); tools/reduce_round.sh turns them into profiles/.
set -euo pipefail
TAG="${1:-r05}"
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}"
X="gpurun_out/${TAG}x"
mkdir -p "$X"
bash tools/collect_profiles.sh "$TAG" all > "$X/collect_profiles.log" 2>&1
bash tools/collect_step_traffic.sh "$TAG" > "$X/collect_traffic.log" 2>&1
python bench.py > "$X/bench_default.json" 2> "$X/bench_default.err"
python tools/layer_table.py > "$X/layer_table.log" 2>&1
python tools/run_configs.py > "$X/configs.log" 2>&1
python bench.py --config cfg3 > "$X/bench_cfg3.json" 2> "$X/bench_cfg3.err"
python bench.py --config cfg5 > "$X/bench_cfg5.json" 2> "$X/bench_cfg5.err"
python bench.py --fp8 --no-cpu-baseline --no-mdct > "$X/bench_fp8.json" 2> "$X/bench_fp8.err"
python bench.py --fp16-storage --no-cpu-baseline --no-mdct > "$X/bench_fp16_storage.json" 2> "$X/bench_fp16_storage.err"
P2PHD_REHEARSE_RCCL=1 python bench.py --wire-bf16 --no-cpu-baseline --no-mdct > "$X/bench_rccl_rehearsal.json" 2> "$X/bench_rccl_rehearsal.err"
for t in ab_halo ab_cls_skip ab_march ab_dfirst ab_dlast; do python tools/$t.py > "$X/$t.log" 2>&1; done
python tools/stress_determinism.py 30 > "$X/determinism_stress.log" 2>&1
python tools/soak.py 200 > "$X/soak.log" 2>&1
tail -n 3 "$X"/*.log | cut -c1-300
