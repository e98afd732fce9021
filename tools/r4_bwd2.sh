#!/bin/bash
set -o pipefail
O=gpurun_out/r4_bwd2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_segments.py -x -q -m gpu > $O/pytest.txt 2>&1 || (tail -30 $O/pytest.txt; exit 1)
tail -2 $O/pytest.txt
bash tools/ab_c3.sh
