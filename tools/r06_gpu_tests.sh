O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.txt 2>&1
rc=$?
tail -15 $O/pytest_gpu_full.txt
exit $rc
