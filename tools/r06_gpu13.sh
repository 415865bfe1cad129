set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
timeout -k 10 300 python -m pytest tests/test_gpu_chol_micro.py tests/test_golden_fixtures.py -m gpu -x -q -s > $O/pytest_new.txt 2>&1 || { tail -30 $O/pytest_new.txt; exit 1; }
grep -i "passed\|device vs oracle\|SCVX_CHOL" $O/pytest_new.txt | tail -8
timeout -k 10 600 python bench.py > $O/bench_B8192_final.json 2> $O/bench_B8192_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_B8192_final.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'cold', d['cold_start_only']['value'], 'K4 ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])
print('traffic model matches', d['roofline']['traffic_model']['calibration_matches_running_library'], d['roofline']['traffic'])
print('1e-5', {k:v for k,v in d.get('value_at_traj_linf_1e-5',{}).items() if k in ('value','failed_steps','retries')})
print('batch32', {k:v for k,v in d['traj_linf_vs_oracle_batch32'].items() if k!='note'})
PY
