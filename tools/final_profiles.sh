#!/bin/bash
# The round's LAST act on the GPU: profiles of the tree as it is, stamped with the hash of the library sources.
#   bash tools/final_profiles.sh r05          (GPU box, from the repo root; ~4 min)
# writes
#   profiles/<tag>_kernel_stats_B8192.csv     rocprofv3 --kernel-trace --stats of tools/pmc_period.py (2 warm-up + 14 solve_steps of the bench mix)
#   profiles/<tag>_pmc_mix_B8192.json         FETCH_SIZE / WRITE_SIZE passes (separate runs) of the same command, per kernel
#   profiles/<tag>_k4_traffic_model.json      bytes per interior-point iteration of socp_kernel = (2 x FETCH + WRITE) / the iterations those
#                                             launches executed, WITH lib_source_hash: bench.py prints it and flags a mismatch with the running tree
# (gpurun_out/ holds the raw directories; profiles/ is what is committed.)
set -e
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
HASH=$(python3 tools/lib_hash.py | cut -d' ' -f1)
mkdir -p gpurun_out profiles
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 tools/pmc_period.py > gpurun_out/${TAG}_stats.log 2>&1
echo "kernel-trace pass done"
for G in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_${G} -- python3 tools/pmc_period.py > gpurun_out/${TAG}_pmc_${G}.log 2>&1
  echo "pmc pass $G done"
done
python3 tools/final_profiles_summarise.py $TAG $HASH
