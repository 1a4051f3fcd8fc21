cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2t
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2t/fetch -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-mdct > gpurun_out/r2t/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2t/write -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-mdct > gpurun_out/r2t/write.log 2>&1
ls gpurun_out/r2t/*/*/*counter_collection.csv; tail -1 gpurun_out/r2t/fetch.log | cut -c1-200
