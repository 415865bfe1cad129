#!/bin/bash
# Instruction-cache and issue counters of the conic kernel over the bench mix (two rocprofv3 --pmc passes of tools/pmc_period.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
L=${1:-successiveconvexification_amd/libscvx_hip.so}
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/icache_a -- python3 tools/pmc_period.py $L > gpurun_out/icache_a.log 2>&1
echo "pass a done"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/icache_b -- python3 tools/pmc_period.py $L > gpurun_out/icache_b.log 2>&1
echo "pass b done"
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/icache_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "socp" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.4g" % tot[k])
if tot.get("SQC_ICACHE_REQ"): print("icache miss rate", tot["SQC_ICACHE_MISSES"] / tot["SQC_ICACHE_REQ"])
if tot.get("SQ_WAVE_CYCLES"): print("wait_inst_any / wave_cycles", tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"], " wait_any", tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"], " active_inst_any", tot.get("SQ_ACTIVE_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"])
PY
