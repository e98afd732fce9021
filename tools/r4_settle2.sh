#!/bin/bash
for rep in 1 2; do
for ms in 0 100 400; do
GSR_BENCH_SETTLE_COPY_MS=$ms python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('copy_ms=$ms', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'], 'clk', d['device_clock_ghz_measured'], d['device_clock_ghz_after'])"
done
done
for w in 5 50 200; do
python bench.py --gpus 1 --steps 20 --warmup $w --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('warmup=$w steps=20', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'])"
done
