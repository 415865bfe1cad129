set -e
O=gpurun_out/r06; mkdir -p $O
rm -f $O/k4_sections_tw2.txt
for L in variants/libscvx_prof_tw2off.so variants/libscvx_hip_prof.so; do
  echo "== B=1024 $L" >> $O/k4_sections_tw2.txt
  timeout -k 10 120 python tools/prof_ipm.py 1024 $L >> $O/k4_sections_tw2.txt 2>&1
done
grep "== B\|chol loop\|border solves\|TOTAL\|wavefront [0-3]:\|newton\|S_solve  \|Et_apply\|residuals" $O/k4_sections_tw2.txt
