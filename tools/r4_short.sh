#!/bin/bash
# the driver's command (20 steps after 5 warm-up steps) three times, then a long run, on one box
set -e -o pipefail
O=gpurun_out/r4_short
mkdir -p $O
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/short$i.json 2> $O/short$i.err; done
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline > $O/long1.json 2> $O/long1.err
for f in short1 short2 short3 long1; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["step_ms"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["roofline"]["avg_launch_ms"], d["device_clock_ghz_first"], d["device_clock_ghz_measured"], d["device_clock_ghz_after"], d["clock_settle_ms"], d["roofline"].get("valu_issue"))
PY
done
