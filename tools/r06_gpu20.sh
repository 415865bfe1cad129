O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full3.txt 2>&1
rc=$?
tail -5 $O/pytest_gpu_full3.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/k4_fuzz_device.py --n 40 --seed 1 --waves 2 > $O/k4_fuzz_device_w2.md 2>&1
tail -4 $O/k4_fuzz_device_w2.md
timeout -k 10 400 python tools/k4_fuzz_device.py --n 24 --seed 7 --waves 2 --fins > $O/k4_fuzz_device_w2_fins.md 2>&1
tail -4 $O/k4_fuzz_device_w2_fins.md
