#!/usr/bin/env python3
"""Merge the per-kernel json files tools/fw_pmc.sh writes (one per precision mode) into the file bench.py reads:
    python tools/merge_pmc.py profiles/r03_pmc_frame_kernel.json gpurun_out/a_pmc_frame_kernel.json gpurun_out/b_..."""
import json
import sys

out, srcs = sys.argv[1], sys.argv[2:]
merged = None
for f in srcs:
    d = json.load(open(f))
    rec = d.pop("frame_fused")
    if merged is None:
        merged = d
        merged["kernels"] = {}
    assert merged["_config"] == d["_config"] and merged["_envs"] == d["_envs"]
    merged["kernels"][rec["kernel"]] = rec
json.dump(merged, open(out, "w"), indent=1)
print(out, list(merged["kernels"]))
