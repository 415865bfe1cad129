set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python tools/bench_k1.py variants/libscvx_r6c.so successiveconvexification_amd/libscvx_hip.so > $O/k1_nb_exo.txt 2>&1
grep -v amdgpu.ids $O/k1_nb_exo.txt
timeout -k 10 600 python -m pytest tests/test_gpu_discretize.py tests/test_golden_fixtures.py -m gpu -x -q > $O/pytest_k1.txt 2>&1 || { tail -30 $O/pytest_k1.txt; exit 1; }
tail -3 $O/pytest_k1.txt
