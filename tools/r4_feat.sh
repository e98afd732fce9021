#!/bin/bash
set -e -o pipefail
O=gpurun_out/r4_feat
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_multi.py tests/test_gpu_render.py tests/test_gpu_segments.py tests/test_gpu_guardband.py -x -q -m gpu > $O/pytest.txt 2>&1 || (tail -40 $O/pytest.txt; exit 1)
tail -2 $O/pytest.txt
python tools/render_stage_ab.py > $O/stage.txt 2>&1; tail -1 $O/stage.txt
bash tools/kstats_render.sh r4_feat > $O/kstats.txt 2>&1; head -24 $O/kstats.txt
