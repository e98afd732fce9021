#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...>   (runs the probe under rocprofv3 --pmc; one pass per call)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python $GRAFT_REPO_ROOT/tools/probe.py --iters 3 --waves 4 > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
echo "pmc $out rc=$?"
