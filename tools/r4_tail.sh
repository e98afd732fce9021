#!/bin/bash
for rep in 1 2; do
for t in 0 2 4 6 8; do
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra --tune blend_tail_cut=$t 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tail_cut=$t', d['value'], 'fps', d['ms_per_step'], 'ms', {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done
done
