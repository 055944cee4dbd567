for m in off on off on; do python bench.py --no-cpu-baseline --split-streams $m --no-stage-timing > gpurun_out/b_$m.json 2>/dev/null && python -c "
import json; d=json.load(open('gpurun_out/b_$m.json')); print('$m', round(d['value'],1), round(d['ms_per_step'],4), d['config']['repeats']['ms_per_step_min'])"; done
