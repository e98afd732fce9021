#!/bin/bash
for rep in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('driver cmd', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'], 'clk', d['device_clock_ghz_measured'], d['device_clock_ghz_after'])"
done
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('long', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'])"
