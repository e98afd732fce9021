#!/bin/bash
# round 4, first GPU call: (a) does the driver's short command (20 steps after 5 warm-up steps) read differently from a long run on the
# same box?  (b) the existing one-wave-per-tile instantiations (four pixels per lane) as a data point for the whole-tile census
set -e -o pipefail
O=gpurun_out/r4_base
mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/short1.json 2> $O/short1.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/short2.json 2> $O/short2.err
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline > $O/long1.json 2> $O/long1.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/short3.json 2> $O/short3.err
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune blend_bwd_waves=1 --tune blend_bwd_reduce=0 > $O/bwd_w1.json 2> $O/bwd_w1.err
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune blend_fwd_waves=1 > $O/fwd_w1.json 2> $O/fwd_w1.err
python bench.py --steps 300 --warmup 50 --no-extra --no-cpu-baseline --tune blend_bwd_waves=2 --tune blend_bwd_reduce=0 > $O/bwd_w2.json 2> $O/bwd_w2.err
for f in short1 short2 long1 short3 bwd_w1 fwd_w1 bwd_w2; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["step_ms"], {k:round(v,4) for k,v in d["stage_ms"].items()}, d["roofline"]["avg_launch_ms"])
PY
done
