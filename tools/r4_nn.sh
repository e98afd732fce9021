#!/bin/bash
set -o pipefail
O=gpurun_out/r4_nn
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_lbs.py tests/test_gpu_render.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
bash tools/kstats_render.sh r4_nn > $O/kstats.txt 2>&1; head -12 $O/kstats.txt; grep -n "nn_cache\|lbs_forward" $O/kstats.txt
