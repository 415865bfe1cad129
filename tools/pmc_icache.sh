#!/bin/bash
# Instruction-cache counters for the bench workload (one extra rocprofv3 --pmc pass; see tools/pmc_passes.sh).
set -e
TAG=${1:-pmc}
B=${2:-8192}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/${TAG}_ICACHE -- python3 bench.py --steps 1 --warmup 0 --batch $B --no-cpu-baseline --no-traj-check --no-k1-sweep > gpurun_out/${TAG}_ICACHE.log 2>&1
echo "pass ICACHE done"
