O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full2.txt 2>&1
rc=$?
tail -6 $O/pytest_gpu_full2.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/k4_fuzz_device.py --n 40 --seed 1 > $O/k4_fuzz_device_seed1_resid.md 2>&1
tail -4 $O/k4_fuzz_device_seed1_resid.md
timeout -k 10 400 python tools/k4_fuzz_device.py --n 40 --seed 1 --retries 0 > $O/k4_fuzz_device_seed1_single_resid.md 2>&1
tail -4 $O/k4_fuzz_device_seed1_single_resid.md
