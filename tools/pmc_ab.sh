#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes (separate rocprofv3 runs, MI355X_MICROARCH.md) of tools/pmc_period.py for each library variant:
#   bash tools/pmc_ab.sh TAG "lib1.so [cold]" "lib2.so" ...     ->  gpurun_out/TAG_<n>_{FETCH_SIZE,WRITE_SIZE}/  + .log
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
n=0
for V in "$@"; do
  for G in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${n}_${G} -- python3 tools/pmc_period.py $V > gpurun_out/${TAG}_${n}_${G}.log 2>&1
    echo "pass $n ($V) $G done"
  done
  n=$((n+1))
done
