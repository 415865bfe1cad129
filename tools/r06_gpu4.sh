set -e
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scvx.py -m gpu -x -q -k "executor or two_ended or socp_matches or horizons or tail" > $O/pytest_twisted.txt 2>&1 || { tail -40 $O/pytest_twisted.txt; exit 1; }
tail -3 $O/pytest_twisted.txt
L="variants/libscvx_r6a.so successiveconvexification_amd/libscvx_hip.so"
B=512 REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_tw_B512.txt 2>&1
cat $O/ab_tw_B512.txt
B=256 REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_tw_B256.txt 2>&1
cat $O/ab_tw_B256.txt
B=1 REPS=3 timeout -k 10 200 python tools/ab_mix.py $L > $O/ab_tw_B1.txt 2>&1
cat $O/ab_tw_B1.txt
