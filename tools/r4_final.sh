#!/bin/bash
# round-4 record on one box: profile_round (kernel stats + PMC passes), the render() frame's kernel table, the driver's command, the default command
tag=${1:-r4b}
set -o pipefail
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1; cat gpurun_out/${tag}_profile.log
KEYS=fused bash tools/kstats_render.sh ${tag}_render > gpurun_out/${tag}_render_kernels_fused_loss.txt 2>&1; head -5 gpurun_out/${tag}_render_kernels_fused_loss.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_cmd.json 2> gpurun_out/${tag}_bench_driver_cmd.err; echo "driver cmd rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/${tag}_bench_driver_cmd2.json 2>/dev/null
python bench.py > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err; echo "default rc=$?"
for f in bench_driver_cmd bench_driver_cmd2 bench_c3; do python - gpurun_out/${tag}_$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["step_ms"], {k:round(v,4) for k,v in d["stage_ms"].items()}, "dom", d["roofline"]["avg_launch_ms"], "clk", d["device_clock_ghz_first"], d["device_clock_ghz_measured"], d["device_clock_ghz_after"])
e=d.get("extra")
if e:
    print("  fwd_only", e["forward_only"]["value"], "C2", e["C2"]["value"], "C5", e["C5"]["value"], "dropin", e["dropin_rasterizer_c3"]["ms_per_step"], e["dropin_rasterizer_c3"].get("blocking_num_rendered_read", {}).get("ms_per_step"), "render eager/graph", e["render_200k"]["eager"]["ms_per_step"], e["render_200k"]["one_graph"].get("ms_per_step"), "torch-loss graph", e["render_200k"]["torch_loss"]["one_graph"].get("ms_per_step"), "c5 parts", e["c5_lbs_dist2"])
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["lbs_project_ms"])
PY
done
