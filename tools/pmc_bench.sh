#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_bench.sh <tag> [bench.py args...]
# Two rocprofv3 --pmc passes (8 SQ counters each) over a short bench.py run; per-kernel averages are printed by
# tools/pmc_kernel.py.  Output: gpurun_out/<tag>_pmcA, _pmcB.
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/${tag}_pmcA -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmcA.log 2>&1 && echo "A ok" &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/${tag}_pmcB -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmcB.log 2>&1 && echo "B ok"
python $root/tools/pmc_kernel.py $out/${tag}_pmcA $out/${tag}_pmcB ${PMC_MATCH:+--match=$PMC_MATCH}
