set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scvx.py -m gpu -x -q -k "executor or two_ended or socp_matches or horizons or tail" > $O/pytest_twisted.txt 2>&1 || { tail -40 $O/pytest_twisted.txt; exit 1; }
tail -3 $O/pytest_twisted.txt
L="variants/libscvx_r6a.so successiveconvexification_amd/libscvx_hip.so"
for B in 1024 512 1; do
B=$B REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_bal_B$B.txt 2>&1
grep -v amdgpu.ids $O/ab_bal_B$B.txt
done
rm -f $O/k4_sections_bal.txt
for B in 1024 512; do
  echo "== B=$B (substitution products on the chain wavefront)" >> $O/k4_sections_bal.txt
  timeout -k 10 120 python tools/prof_ipm.py $B variants/libscvx_hip_prof.so >> $O/k4_sections_bal.txt 2>&1
done
grep "== B\|chol loop\|border solves\|TOTAL\|wavefront [0-3]:" $O/k4_sections_bal.txt
