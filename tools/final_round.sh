#!/bin/bash
# usage (GPU box, repo root): bash tools/final_round.sh <tag>
# The end-of-round evidence in one call: tools/profile_round.sh (rocprofv3 stats / timelines / PMC of the frame kernel), the
# PMC summary copied where bench.py reads `roofline.traffic` from, the default bench line, the line with the driver's
# arguments, and the configs[4] profile.  Everything under gpurun_out/<tag>_*.
R=$PWD; TAG=${1:-r04}
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1 || echo "profile_round failed"
cp gpurun_out/${TAG}_pmc_frame_kernel.json profiles/${TAG}_pmc_frame_kernel.json
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_line_1gpu.json 2> gpurun_out/${TAG}_bench_1gpu.err || echo "bench failed"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line_driver_args.json 2> gpurun_out/${TAG}_bench_driver_args.err || echo "bench (driver args) failed"
bash tools/profile_config5.sh $TAG > gpurun_out/${TAG}_profile_config5.log 2>&1 || echo "profile_config5 failed"
python - <<PY
import json
for n in ("1gpu", "driver_args"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_bench_line_%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, "value %.0f ms/step %.4f no_reset %.4f whole_episode %s frame %.4f frac %.3f alone %.4f traffic %s" % (
            d["value"], d["ms_per_step"], d["ms_per_step_no_reset"], (d.get("whole_episode") or {}).get("value"),
            r["avg_launch_ms"], r["frac"], r["alone"]["avg_launch_ms"], r.get("traffic")))
        for name, c in (d.get("configs") or {}).items():
            print("   ", name, c.get("workload"), "%.0f" % c.get("value", float("nan")))
    except Exception as e:
        print(n, "unreadable:", e)
PY
