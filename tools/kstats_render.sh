#!/bin/bash
# usage (GPU box, repo root): bash tools/kstats_render.sh <tag>  -> per-kernel time of one render() frame (phase-1 loss), 53 frames run
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export PROFILE=1 KEYS=${KEYS:-phase1}   # KEYS=fused: the phase-1 training loss fused with the rasterizer
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_rstats -- python $root/tools/render_bench.py > $out/${tag}_rstats.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/${tag}_rstats/*/*_kernel_stats.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/53/1e3
print(f"GPU time per frame: {tot:.1f} us over {len(rows)} distinct kernels")
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/53/1e3:9.2f} us/frame  {float(r["AverageNs"])/1e3:8.2f} us x{int(r["Calls"])/53:6.1f}  {r["Name"].replace("void ","").replace("gsr::","")[:100]}')
PY
