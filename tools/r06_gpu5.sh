set -e
O=gpurun_out/r06; mkdir -p $O
for B in 512 1; do
  echo "== B=$B (two-ended, t-space border)" >> $O/k4_sections_tw.txt
  timeout -k 10 120 python tools/prof_ipm.py $B variants/libscvx_hip_prof.so >> $O/k4_sections_tw.txt 2>&1
done
cat $O/k4_sections_tw.txt
