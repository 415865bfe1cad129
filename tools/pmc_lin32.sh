#!/bin/bash
# HBM traffic of the conic solve with the derivative tiles in double and in float: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes of tools/lin32_step.py.   bash tools/pmc_lin32.sh <tag> ; python tools/pmc_summarise.py <tag> out.json
set -e
TAG=${1:-lin32pmc}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for G in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${G} -- python3 tools/lin32_step.py 8192 > gpurun_out/${TAG}_${G}.log 2>&1
    echo "pass $G done"; tail -2 gpurun_out/${TAG}_${G}.log
done
