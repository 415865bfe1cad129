set -e
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 bash tools/final_profiles.sh r06 > $O/final_profiles.log 2>&1 || { tail -30 $O/final_profiles.log; exit 1; }
tail -5 $O/final_profiles.log
echo progress: profiles done
timeout -k 10 300 python bench.py --aero --batch 256 > $O/bench_aero_B256.json 2> $O/bench_aero_B256.err
echo progress: aero done
timeout -k 10 900 python bench.py --config5 > $O/bench_config5.json 2> $O/bench_config5.err
python - <<'PY'
import json
for f in ('bench_aero_B256','bench_config5'):
    d=json.loads(open('gpurun_out/r06/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['workload'][:100])
PY
