#!/bin/bash
# per-kernel time of the skinning-offset network's fused forward + backward (tools/mlp_bench.py) -- GPU box, repo root
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mlp_stats -- python $root/tools/mlp_bench.py > $out/mlp_stats.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/mlp_stats/*/*_kernel_stats.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us avg x{r["Calls"]:>5}  {r["Name"].replace("void ","").replace("gsr::","")[:110]}')
PY
grep "^P=" $out/mlp_stats.log
