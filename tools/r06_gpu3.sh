set -e
O=gpurun_out/r06
mkdir -p $O
for B in 8192 1024 512 1; do
  for L in variants/libscvx_prof0.so variants/libscvx_hip_prof.so; do
    echo "== B=$B $L" >> $O/k4_sections.txt
    timeout -k 10 120 python tools/prof_ipm.py $B $L >> $O/k4_sections.txt 2>&1
  done
done
cat $O/k4_sections.txt
