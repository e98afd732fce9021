#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_modes.sh <tag> "<tune A>" "<tune B>" ...  -- full bench line (C3 + extras) per tuning set, interleaved twice
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
for rep in 1 2; do
  i=0
  for t in "$@"; do
    i=$((i+1))
    args=""
    for kv in $t; do args="$args --tune $kv"; done
    python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline $args > $out/${tag}_${i}_r${rep}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$out/${tag}_${i}_r${rep}.json"))
e = d["extra"]
print("[$t] C3", d["value"], "fps", d["ms_per_step"], "ms | fwd-only", e["forward_only"]["value"], "| C2", e["C2"]["value"], "| C5", e["C5"]["value"],
      "| dropin", e["dropin_rasterizer_c3"]["value"], "| render eager", e["render_200k"]["eager"]["ms_per_step"], "graph", e["render_200k"].get("one_graph", {}).get("ms_per_step"),
      "| stages", {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()})
PY
  done
done
