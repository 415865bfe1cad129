#!/bin/bash
# Kernel trace + PMC passes for K0 (the 3-DoF initialiser), B = 8192, one counter group per rocprofv3 run (never with
# --sys-trace / --hip-trace).  Usage on the GPU box: bash tools/pmc_k0.sh <tag>
TAG=$1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_STATS -- python3 tools/threedof_bench.py --B 8192 --reps 5 > gpurun_out/${TAG}_STATS.log 2>&1 || exit 1
# (a TCC / SQ group pass of this persistent kernel ran into the box's 7-minute silence guard in round 2: FETCH / WRITE only)
for SPEC in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE"; do
    NAME=${SPEC%%:*}; G=${SPEC#*:}
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${NAME} -- python3 tools/threedof_bench.py --B 8192 --reps 1 > gpurun_out/${TAG}_${NAME}.log 2>&1 || exit 1
done
