set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python tools/failed_steps_at_tol.py 3e-10 0 5 7 > $O/failed_steps_3e-10.md 2>&1
cat $O/failed_steps_3e-10.md
timeout -k 10 300 python tools/failed_steps_at_tol.py 1e-10 5 7 > $O/failed_steps_1e-10.md 2>&1
cat $O/failed_steps_1e-10.md
