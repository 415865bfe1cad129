#!/bin/bash
# Round-3 counter passes (one counter group per rocprofv3 run; FETCH_SIZE and WRITE_SIZE separately, MI355X_MICROARCH.md):
# the default bench workload (exo: socp_kernel + linearize_pcp_kernel<false>) and the aero model at the same batch
# (linearize_pcp_kernel<true>).  Usage on the GPU box:  bash tools/pmc_r03.sh <tag>
# Outputs gpurun_out/<tag>_<model>_<GROUP>/ ; summarise with tools/pmc_summarise.py <tag>_<model> <out.json>.
set -e
TAG=${1:-r03pmc}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for MODEL in exo aero; do
  EXTRA=""; [ "$MODEL" = "aero" ] && EXTRA="--aero"
  for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
    NAME=$(echo $G | cut -d' ' -f1)
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_${MODEL}_${NAME} -- python3 bench.py --steps 1 --warmup 0 --batch 8192 $EXTRA --no-cpu-baseline --no-traj-check --no-k1-sweep > gpurun_out/${TAG}_${MODEL}_${NAME}.log 2>&1
    echo "pass $MODEL $NAME done"
  done
done
