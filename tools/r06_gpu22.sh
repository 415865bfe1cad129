set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python tools/bench_k1.py variants/libscvx_r6c.so successiveconvexification_amd/libscvx_hip.so > $O/k1_nb1_check.txt 2>&1
grep -v amdgpu.ids $O/k1_nb1_check.txt
