#!/bin/bash
# Instruction-mix / cache / LDS counters of the conic kernel over the bench mix: rocprofv3 --pmc passes of tools/pmc_period.py, a few counters each
#   bash tools/pmc_mix_counters.sh [lib.so]   ->  gpurun_out/mixc_*  + a summary on stdout
set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=${1:-successiveconvexification_amd/libscvx_hip.so}
n=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum"; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/mixc_$n -- python3 tools/pmc_period.py $L > gpurun_out/mixc_$n.log 2>&1 || echo "pass $n failed"
  echo "pass $n done"
  n=$((n+1))
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/mixc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "socp" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.4g" % tot[k])
PY
