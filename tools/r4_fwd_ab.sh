#!/bin/bash
set -e -o pipefail
O=gpurun_out/r4_fwd
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_segments.py -x -q -m gpu > $O/pytest.txt 2>&1 || (tail -30 $O/pytest.txt; exit 1)
for i in 1 2; do
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune blend_layout=2 > $O/old$i.json 2> $O/old$i.err
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline > $O/new$i.json 2> $O/new$i.err
done
for f in old1 new1 old2 new2; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["step_ms"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["roofline"]["avg_launch_ms"])
PY
done
