#!/bin/bash
# usage (GPU box, repo root): bash tools/vpr_kstats.sh -> per-kernel GPU time of one `bench.py --workload render` step (stand-in decoders on)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/vpr_rstats -- python $root/bench.py --workload render --steps 40 --warmup 10 > $out/vpr_rstats.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/vpr_rstats/*/*_kernel_stats.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:-float(r["TotalDurationNs"]))
n=50
tot=sum(float(r["TotalDurationNs"]) for r in rows)/n/1e3
print(f"GPU time per step (approx, {n} steps): {tot:.1f} us over {len(rows)} distinct kernels")
for r in rows[:30]:
    print(f'{float(r["TotalDurationNs"])/n/1e3:9.2f} us/step  {float(r["AverageNs"])/1e3:8.2f} us x{int(r["Calls"])/n:6.1f}  {r["Name"].replace("void ","").replace("gsr::","")[:110]}')
PY
