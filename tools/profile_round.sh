#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Collects, for the default bench command: the rocprofv3 kernel-trace stats, and separate PMC passes for HBM traffic
# (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) and wave / VALU activity.  Output: gpurun_out/<tag>_*.
set -o pipefail
tag=$1; shift   # remaining arguments go to bench.py (e.g. --tune tile_order=2)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra "$@" > $out/${tag}_stats.log 2>&1 && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmc_fetch.log 2>&1 && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmc_write.log 2>&1 && echo "write ok" &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/${tag}_pmc_sq -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmc_sq.log 2>&1 && echo "sq ok" &&
# (round 4) the LDS array: busy cycles, bank-conflict cycles, instruction count, wave-cycles waiting on it
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $out/${tag}_pmc_lds -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/${tag}_pmc_lds.log 2>&1 && echo "lds ok"
