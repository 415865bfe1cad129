#!/bin/bash
# PMC passes for the bench workload, one counter group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE separately, as
# MI355X_MICROARCH.md prescribes).  Usage on the GPU box:  bash tools/pmc_passes.sh <tag> [batch]
# Outputs gpurun_out/<tag>_<GROUP>/ ; summarise with tools/pmc_summarise.py <tag> <out.json>.
set -e
TAG=${1:-pmc}
B=${2:-8192}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum"; do
    NAME=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${NAME} -- python3 bench.py --steps 1 --warmup 0 --batch $B --no-cpu-baseline --no-traj-check --no-k1-sweep > gpurun_out/${TAG}_${NAME}.log 2>&1
    echo "pass $NAME done"
done
