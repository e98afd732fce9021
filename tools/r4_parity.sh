#!/bin/bash
set -o pipefail
O=gpurun_out/r4_parity
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "benched_session or render_bench_scene" > $O/new_tests.txt 2>&1; tail -25 $O/new_tests.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parallel_render.py tests/test_host_api.py -x -q -k "replaced_leaf or dropin" > $O/leaf.txt 2>&1; tail -5 $O/leaf.txt
GSR_RENDER_GRAD_TOL=1e-4 timeout -k 10 600 python -m pytest tests/test_gpu_render.py -q -m gpu -k "parameter_gradients_match" > $O/tol1e4.txt 2>&1; grep -n "Error\|error\|assert\|passed\|failed" $O/tol1e4.txt | head -20
