set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scvx.py -m gpu -x -q -k "executor or two_ended or socp_matches or horizons or tail" > $O/pytest_tw2.txt 2>&1 || { tail -40 $O/pytest_tw2.txt; exit 1; }
tail -3 $O/pytest_tw2.txt
L="variants/libscvx_r6c.so successiveconvexification_amd/libscvx_hip.so"
for B in 1024 768; do
B=$B REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_tw2_B$B.txt 2>&1
grep -v amdgpu.ids $O/ab_tw2_B$B.txt
done
