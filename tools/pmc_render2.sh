#!/bin/bash
# usage: bash tools/pmc_render2.sh <tag>  -> latency-related SQ counters of the blend kernels in the render() frame
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ONLY=1 KEYS=phase1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVES SQ_ACCUM_PREV_HIRES --kernel-trace --output-format csv -d $out/${tag}_rpmcC -- python $root/tools/render_bench.py > $out/${tag}_rpmcC.log 2>&1 && echo "C ok"
python $root/tools/pmc_kernel.py $out/${tag}_rpmcC --match=blend
