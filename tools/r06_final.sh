# Round 6, the last act on the GPU: profiles of the tree as it is (stamped with the library source hash), the bench lines, the sweeps.
set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 bash tools/final_profiles.sh r06 > $O/final_profiles.log 2>&1 || { tail -30 $O/final_profiles.log; exit 1; }
tail -3 $O/final_profiles.log
echo progress: profiles done
timeout -k 10 600 python bench.py > $O/bench_B8192_final.json 2> $O/bench_B8192_final.err
echo progress: bench done
timeout -k 10 300 python bench.py --aero --batch 256 > $O/bench_aero_B256.json 2> $O/bench_aero_B256.err
echo progress: aero done
timeout -k 10 900 python bench.py --config5 > $O/bench_config5.json 2> $O/bench_config5.err
echo progress: config5 done
timeout -k 10 400 python tools/bsweep_mix.py > $O/bsweep_final.md 2>&1
timeout -k 10 200 python tools/cold_step_latency.py > $O/cold_latency_final.md 2>&1
python - <<'PY'
import json
for f in ('bench_B8192_final','bench_aero_B256','bench_config5'):
    d=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'cold', round(d['cold_start_only']['value']))
d=json.loads(open('gpurun_out/r06/bench_B8192_final.json').read().strip().splitlines()[-1])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_model']['calibration_matches_running_library'], d['roofline']['traffic_model']['bytes_per_ipm_iteration'])
print('1e-5', {k:v for k,v in d.get('value_at_traj_linf_1e-5',{}).items() if k in ('value','failed_steps','retries')})
print('batch32', {k:v for k,v in d['traj_linf_vs_oracle_batch32'].items() if k!='note'})
print('sample', d['traj_linf_vs_oracle'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['accepted_steps_per_s'], d['solve_problems_per_s'], d['f32_linearization']['value'])
PY
sed -n '/^| B per GPU/,$p' $O/bsweep_final.md
grep "^|" $O/cold_latency_final.md
