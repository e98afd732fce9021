#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_tile_order.sh <tag>   -- A/B of Options::tile_order 1 / 2 / 3 at C3: bench time and the
# memory-side traffic of the two blend kernels (separate FETCH_SIZE / WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes)
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 1 2 3 1 2 3; do
  python $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --tune tile_order=$m > $out/${tag}_bench_o$m.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$out/${tag}_bench_o$m.json"))
print("tile_order=$m", d["value"], "frames/s", d["ms_per_step"], "ms", d["stage_ms"])
PY
done
for m in 1 2 3; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_o${m}_fetch -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --tune tile_order=$m > $out/${tag}_o${m}_fetch.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_o${m}_write -- python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --tune tile_order=$m > $out/${tag}_o${m}_write.log 2>&1 &&
  echo "pmc tile_order=$m ok"
done
python $root/tools/pmc_traffic.py $out/${tag}_o1 $out/${tag}_o2 $out/${tag}_o3
