#!/bin/bash
# usage (GPU box, repo root): bash tools/kstats.sh <tag> [bench.py args...]  -> per-kernel average durations (rocprofv3 --kernel-trace --stats)
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra "$@" > $out/${tag}_stats.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/${tag}_stats/*/*_kernel_stats.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{float(r["AverageNs"])/1e3:9.2f} us x{r["Calls"]:>5}  {r["Name"].replace("void ","").replace("gsr::","")[:90]}')
PY
