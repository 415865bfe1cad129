#!/bin/bash
# Round-3 final counter passes over the bench's own timed mix (2 warm-up + 14 timed solve_steps of the default workload), one counter
# group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE separately, MI355X_MICROARCH.md).  Usage on the GPU box: bash tools/pmc_r03_mix.sh <tag>
# Outputs gpurun_out/<tag>_exo_<GROUP>/ ; summarise with tools/pmc_summarise.py <tag>_exo <out.json>.
set -e
TAG=${1:-r03mix}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $G | cut -d' ' -f1)
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_exo_${NAME} -- python3 bench.py --steps 14 --warmup 2 --batch 8192 --no-cpu-baseline --no-traj-check --no-k1-sweep > gpurun_out/${TAG}_exo_${NAME}.log 2>&1
  echo "pass exo $NAME done"
done
