#!/bin/bash
# Kernel trace + PMC passes for K0 (the 3-DoF initialiser), B = 8192, one counter group per rocprofv3 run (never with
# --sys-trace / --hip-trace).  Usage on the GPU box: bash tools/pmc_k0.sh <tag>
TAG=$1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_STATS -- python3 tools/threedof_bench.py --B 8192 --reps 5 > gpurun_out/${TAG}_STATS.log 2>&1 || exit 1
for SPEC in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE" "TCCHIT:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    NAME=${SPEC%%:*}; G=${SPEC#*:}
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${NAME} -- python3 tools/threedof_bench.py --B 8192 --reps 1 > gpurun_out/${TAG}_${NAME}.log 2>&1 || exit 1
done
python3 tools/pmc_summarise.py ${TAG} > gpurun_out/${TAG}_summary.json
