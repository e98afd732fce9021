#!/bin/bash
# NOTE: a record of how profiles/r4_group_fetch_experiment.txt was measured.  It needs the experiment patch of that commit's parent
# work tree (blend_bwd_reduce = 5 / 6 selected the per-survivor / one-block forms); the tree no longer accepts those knob values.
# round 4: the plain backward with a group's four survivors fetched together -- parity (default = branchy form, then the one-block
# form through GSR_TEST_TUNING), then same-box A/B of blend_bwd_reduce = 5 (per-survivor form) / 3 (default) / 6 (one block),
# and the render() frame old tree vs new (five-wave 18-channel forward)
set -e -o pipefail
O=gpurun_out/r4_group
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_segments.py -x -q -m gpu > $O/pytest_default.txt 2>&1
GSR_TEST_TUNING=blend_bwd_reduce=6 timeout -k 10 600 python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_segments.py -x -q -m gpu > $O/pytest_oneblock.txt 2>&1
for i in 1 2; do
for m in 5 3 6; do
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune blend_bwd_reduce=$m > $O/m${m}_$i.json 2> $O/m${m}_$i.err
done
done
for f in m5_1 m3_1 m6_1 m5_2 m3_2 m6_2; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["roofline"]["avg_launch_ms"])
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_multi.py tests/test_gpu_render.py -x -q -m gpu > $O/pytest_multi.txt 2>&1
bash tools/r4_feat_ab.sh
