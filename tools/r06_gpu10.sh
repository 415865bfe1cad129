set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python tools/k4_fuzz_device.py --n 40 --seed 1 > $O/k4_fuzz_device_seed1.md 2>&1
tail -4 $O/k4_fuzz_device_seed1.md
timeout -k 10 400 python tools/k4_fuzz_device.py --n 40 --seed 1 --retries 0 > $O/k4_fuzz_device_seed1_single.md 2>&1
tail -4 $O/k4_fuzz_device_seed1_single.md
timeout -k 10 400 python tools/k4_fuzz_device.py --n 24 --seed 7 --fins --retries 0 > $O/k4_fuzz_device_seed7_fins_single.md 2>&1
tail -4 $O/k4_fuzz_device_seed7_fins_single.md
timeout -k 10 400 python tools/bsweep_mix.py > $O/bsweep_final.md 2>&1
tail -22 $O/bsweep_final.md
timeout -k 10 200 python tools/cold_step_latency.py > $O/cold_latency_final.md 2>&1
cat $O/cold_latency_final.md
