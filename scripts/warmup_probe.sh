# how the length of the warm-up changes a 20-step measurement (the driver runs --steps 20 --warmup 5)
for w in 5 60 300; do timeout -k 10 100 python3 bench.py --steps 20 --warmup $w --no-e2e --no-cpu-baseline > gpurun_out/w.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/w.json')); print('warmup', $w, 'ms/step %.4f' % d['ms_per_step'], 'quantile', d['config']['quantile_frames'], 'redone', d['config']['redone_frames'])"; done
for w in 5 60; do timeout -k 10 100 python3 bench.py --steps 20 --warmup $w --no-e2e --no-cpu-baseline --jump-every 0 > gpurun_out/w.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/w.json')); print('no box misses, warmup', $w, 'ms/step %.4f' % d['ms_per_step'])"; done
