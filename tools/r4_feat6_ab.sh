#!/bin/bash
# same-box A/B of the 18-channel backward instances in the render() frame: committed tree (tools/scratch/old) vs working tree.
# per-kernel GPU time with all six triples live (KEYS=all -> blend_backward_features_kernel_6) and with the phase-1 images (-> _2)
base=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for which in old new; do
  dir=$base; [ $which = old ] && dir=$base/tools/scratch/old
  for keys in all phase1; do
    out=$base/gpurun_out/f6_${which}_${keys}_$rep
    PROFILE=1 KEYS=$keys rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $dir/tools/render_bench.py > $out.log 2>&1
    python - <<PY
import csv,glob
f=glob.glob("$out/*/*_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    if "blend_backward_features" in r["Name"] or "blend_forward_kernel" in r["Name"]:
        print("$which $keys", r["Name"].split("(")[0].replace("void gsr::",""), f'{float(r["AverageNs"])/1e3:.1f} us x{r["Calls"]}')
PY
  done
done
done
