set -e
mkdir -p gpurun_out/r06 /tmp/b
O=gpurun_out/r06
for v in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSCVX_CHOL_DPP=$v -Iinclude -Isuccessiveconvexification_amd/csrc -o /tmp/b/chol_$v tools/micro/chol_dpp_ab.hip; done
for v in 0 1 2; do timeout -k 10 60 /tmp/b/chol_$v 2048; timeout -k 10 60 /tmp/b/chol_$v 64; done > $O/chol_dpp_micro.txt 2>&1
cat $O/chol_dpp_micro.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/b/sps tools/micro/stream_phase_shapes.hip
timeout -k 10 200 /tmp/b/sps 8192 200 > $O/stream_phase_shapes.txt 2>&1
tail -5 $O/stream_phase_shapes.txt
L="variants/libscvx_chol0.so successiveconvexification_amd/libscvx_hip.so variants/libscvx_chol1.so"
timeout -k 10 300 python tools/ab_mix.py $L > $O/ab_chol_B8192.txt 2>&1
cat $O/ab_chol_B8192.txt
B=1024 REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_chol_B1024.txt 2>&1
cat $O/ab_chol_B1024.txt
B=512 REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_chol_B512.txt 2>&1
cat $O/ab_chol_B512.txt
timeout -k 10 500 python -m pytest tests/test_gpu_scvx.py -m gpu -x -q -k "executor or socp or twisted or two_ended" > $O/pytest_k4_subset.txt 2>&1
tail -5 $O/pytest_k4_subset.txt
