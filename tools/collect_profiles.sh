cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2p/step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-mdct > gpurun_out/r2p/step.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2p/fetch -- python3 tools/pmc_targets.py > gpurun_out/r2p/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2p/write -- python3 tools/pmc_targets.py > gpurun_out/r2p/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2p/mfma -- python3 tools/pmc_targets.py > gpurun_out/r2p/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d gpurun_out/r2p/sq -- python3 tools/pmc_targets.py > gpurun_out/r2p/sq.log 2>&1
find gpurun_out/r2p -name "*.csv" | head -30; tail -2 gpurun_out/r2p/*.log
python3 bench.py --fp8 --no-cpu-baseline --no-mdct > gpurun_out/r2p/bench_fp8.json 2> gpurun_out/r2p/bench_fp8.err; tail -3 gpurun_out/r2p/bench_fp8.err
