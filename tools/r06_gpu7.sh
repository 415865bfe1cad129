set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_B8192.json 2> $O/bench_B8192.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_B8192.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'cold', d['cold_start_only']['value'], 'K4 ms', d['roofline']['avg_launch_ms'])
print('batch32', d.get('traj_linf_vs_oracle_batch32'))
print('sample', d.get('traj_linf_vs_oracle'))
print('1e-5', {k:v for k,v in d.get('value_at_traj_linf_1e-5',{}).items() if k in ('value','failed_steps')})
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
timeout -k 10 400 python tools/bsweep_mix.py > $O/bsweep.md 2>&1
cat $O/bsweep.md
timeout -k 10 200 python tools/cold_step_latency.py > $O/cold_latency.md 2>&1
cat $O/cold_latency.md
