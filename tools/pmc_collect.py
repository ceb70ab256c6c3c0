#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one directory per pass, counter_collection.csv inside) into a
per-kernel JSON: mean over the last `--last` dispatches of every kernel family, plus the HBM bytes
per launch as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE is in KiB and counts 128-byte
requests at 64 bytes: x2; WRITE_SIZE in KiB is exact)."""
import argparse, csv, glob, json, os, re, collections
FAMILIES = [("frame_fused", r"k_frame_wave"), ("target_finish", r"k_target_finish"),
            ("gemm_nt", r"k_gemm_nt2|k_gemm_nt<"), ("gemm_reduce", r"k_gemm_reduce"),
            ("extrude_gather", r"k_extrude_gather"), ("extrude_scatter", r"k_extrude_scatter"),
            ("dm_shape", r"k_dm_shape_sep"), ("wfs_spot_cog", r"k_wfs_spot"),
            ("target_psf", r"k_target_rows")]
ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+"); ap.add_argument("--last", type=int, default=3)
ap.add_argument("--envs", type=int, default=256); ap.add_argument("--how", default="")
a = ap.parse_args()
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # family -> counter -> [per dispatch]
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # (family, dispatch) -> counter -> sum
        for r in csv.DictReader(open(f)):
            fam = next((n for n, rx in FAMILIES if re.search(rx, r["Kernel_Name"])), None)
            if fam is None:
                continue
            per[(fam, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
        byfam = collections.defaultdict(list)
        for (fam, disp), c in sorted(per.items()):
            byfam[fam].append(c)
        for fam, lst in byfam.items():
            for c in lst[-a.last:]:
                for k, v in c.items():
                    vals[fam][k].append(v)
out = {"_envs": a.envs, "_how": a.how}
for fam, cs in vals.items():
    o = {k: sum(v) / len(v) for k, v in cs.items()}
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_traffic_bytes_per_launch"] = o["FETCH_SIZE"] * 1024 * 2 + o["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o.get("GRBM_GUI_ACTIVE", 0) > 0:
        # same normalisation as the earlier rounds' files: busy cycles / (GUI-active cycles x 128)
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] * 128.0)
    out[fam] = o
print(json.dumps(out, indent=1, sort_keys=True))
