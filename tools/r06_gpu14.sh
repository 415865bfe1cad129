set -e
O=gpurun_out/r06; mkdir -p $O
L="variants/libscvx_r6b.so successiveconvexification_amd/libscvx_hip.so"
timeout -k 10 300 python tools/ab_mix.py $L > $O/ab_resid_B8192.txt 2>&1
grep -v amdgpu.ids $O/ab_resid_B8192.txt
for B in 1024 512; do
B=$B REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_resid_B$B.txt 2>&1
grep -v amdgpu.ids $O/ab_resid_B$B.txt
done
