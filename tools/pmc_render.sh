#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_render.sh <tag> [match]  -> SQ counters of the blend kernels in the render() frame (phase-1 loss)
tag=$1; match=${2:-blend}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ONLY=1 KEYS=phase1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/${tag}_rpmcA -- python $root/tools/render_bench.py > $out/${tag}_rpmcA.log 2>&1 && echo "A ok" &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out/${tag}_rpmcB -- python $root/tools/render_bench.py > $out/${tag}_rpmcB.log 2>&1 && echo "B ok"
python $root/tools/pmc_kernel.py $out/${tag}_rpmcA $out/${tag}_rpmcB --match=$match
