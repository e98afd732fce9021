#!/bin/bash
set -o pipefail
O=gpurun_out/r4_render
mkdir -p $O
python - > $O/render_extra.json 2> $O/render_extra.err <<'PY'
import json, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.getcwd())
import torch
import bench
print(json.dumps(bench.render_extra(torch.device("cuda", 0)), indent=1))
PY
cat $O/render_extra.json; tail -3 $O/render_extra.err
