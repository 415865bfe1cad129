set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_golden_fixtures.py -m gpu -x -q -s -k "aero_B256_full_run or headline_batch" > $O/pytest_fixt.txt 2>&1 || { tail -30 $O/pytest_fixt.txt; exit 1; }
grep "device vs oracle\|passed" $O/pytest_fixt.txt
