mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/gputests_final2.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/s2/gputests_final2.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench_driver2.json 2> gpurun_out/s2/bench_driver2.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s2/bench_driver2.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['traffic'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(' ', k[:60], round(v['value']/1e6,2), v.get('kernel'), v.get('kernel_ms'), v.get('traffic'), (round(v['resident_rollout']['value']/1e6,1), v['resident_rollout'].get('kernel'), v['resident_rollout'].get('traffic')) if 'resident_rollout' in v else (v.get('kernel_ms_per_tick'), v.get('traffic')))
PY
