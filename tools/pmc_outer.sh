set -euo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r3pmc
rm -rf $OUT; mkdir -p $OUT/fetch $OUT/write $OUT/tcc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/one_layer.py 48 96 512 256 3 2 1 0 0 0 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/one_layer.py 48 96 512 256 3 2 1 0 0 0 > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- python3 tools/one_layer.py 48 96 512 256 3 2 1 0 0 0 > $OUT/tcc.log 2>&1
find $OUT -name "*counter_collection.csv"
