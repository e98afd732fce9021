#!/bin/bash
# round 4: the two per-tile sort launches merged into one (knob bucket_sort_merged): parity, then same-build A/B at C3 and in the render() frame
set -e -o pipefail
O=gpurun_out/r4_sortmerge
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sort_knn.py tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_segments.py tests/test_gpu_multi.py tests/test_gpu_render.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -n 2 $O/pytest.txt
for i in 1 2 3; do
for m in 0 1; do
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune bucket_sort_merged=$m 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('merged=$m', d['value'], 'fps', d['ms_per_step'], 'ms', {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
done
done
for i in 1 2 3; do
for m in 0 1; do
python tools/render_stage_ab.py bucket_sort_merged=$m 2>/dev/null | tail -n 1
done
done
