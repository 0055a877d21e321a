mkdir -p gpurun_out/s2
CAT_TIMING_LIB=build/var/pool_timing.so python tools/rollout_phase.py labyrinth 4096 64 > gpurun_out/s2/phase_pool_T64.txt 2>&1
CAT_TIMING_LIB=build/var/pool_timing.so python tools/rollout_phase.py labyrinth 4096 0 > gpurun_out/s2/phase_pool_step.txt 2>&1
cut -c1-90 gpurun_out/s2/phase_pool_T64.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/s2/bench_pool.json
python -c "
import json; d=json.load(open('gpurun_out/s2/bench_pool.json')); print(d['value']/1e6, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms']); 
for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('value','kernel_ms','kernel','kernel_ms_per_tick')} if isinstance(v,dict) else v)
"
