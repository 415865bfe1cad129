set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY --kernel-trace --output-format csv -d gpurun_out/utcl -- python3 tools/pmc_period.py successiveconvexification_amd/libscvx_hip.so > gpurun_out/utcl.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS --kernel-trace --output-format csv -d gpurun_out/utcl2 -- python3 tools/pmc_period.py successiveconvexification_amd/libscvx_hip.so > gpurun_out/utcl2.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/utcl*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "socp" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, "%.4g" % tot[k])
PY
