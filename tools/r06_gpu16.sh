set -e
O=gpurun_out/r06; mkdir -p $O
L="successiveconvexification_amd/libscvx_hip.so variants/libscvx_fresh6.so"
timeout -k 10 300 python tools/ab_mix.py $L > $O/ab_fresh6_B8192.txt 2>&1
grep -v amdgpu.ids $O/ab_fresh6_B8192.txt
