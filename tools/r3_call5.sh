set -o pipefail
timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r3f_tests.log 2>&1; echo "all tests rc=$?"; tail -6 gpurun_out/r3f_tests.log
timeout -k 10 200 python tools/shard_probe.py supersaw > gpurun_out/r3f_shard_ss.txt 2>&1; cat gpurun_out/r3f_shard_ss.txt
t0=$SECONDS
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err || { tail -5 gpurun_out/r3f_bench.err; }
echo "bench wall $((SECONDS - t0)) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
print(json.dumps(d['config'].get('highlights'),indent=0)[:3000])
print(json.dumps(d['roofline'])[:600])
print(json.dumps(d.get('north_star_pes',{}).get('rows'),indent=0)[:3000])
print(json.dumps(d.get('north_star_pes',{}).get('comb_bank_512')))
PY
