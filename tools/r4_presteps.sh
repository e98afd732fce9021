#!/bin/bash
# how many untimed steps of the workload in front of the warm-up does the driver's short command need to read like a long run?
for rep in 1 2 3; do
for n in 0 64 256; do
GSR_BENCH_PRESTEPS=$n python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('presteps=$n', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'], 'clk', d['device_clock_ghz_first'], d['device_clock_ghz_measured'], d['device_clock_ghz_after'])"
done
python bench.py --gpus 1 --steps 300 --warmup 50 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('long run   ', d['value'], 'fps', d['ms_per_step'], 'ms', d['step_ms'], 'dom', d['roofline']['avg_launch_ms'])"
done
