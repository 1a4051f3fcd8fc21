#!/usr/bin/env bash
# Memory-side traffic of one whole (eager) step (2 step-equivalents per pass: 1 warm-up + 1 timed, no probe steps): two separate --pmc passes, fresh output directories (see collect_profiles.sh).
set -euo pipefail
TAG="${1:-r05}"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}"
OUT="gpurun_out/${TAG}t"
rm -rf "$OUT/fetch" "$OUT/write"; mkdir -p "$OUT/fetch" "$OUT/write"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-probes --no-cpu-baseline --no-mdct > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-probes --no-cpu-baseline --no-mdct > "$OUT/write.log" 2>&1
ls "$OUT"/*/*/*counter_collection.csv
tail -n 1 "$OUT/fetch.log" | cut -c1-200
