"""Condense the rocprofv3 --pmc dumps written by tools/pmc_passes.sh into one JSON per round."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
src = f"gpurun_out/pmc_{tag}"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{src}/*/vm/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = "blind_rotate_512_kernel" if "blind_rotate_512_kernel" in name else "key_switch_kernel" if "key_switch_kernel" in name else None
        if key:
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = sum(v) / len(v)  # average per launch
for k, d in res.items():
    if "FETCH_SIZE" in d:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
        # wide coalesced stream, so it is doubled before it is compared with a byte count
        d["hbm_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2 + d.get("WRITE_SIZE", 0.0) * 1024
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d:
        d["valu_active_share_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
out = {"workload": "tools/prof_target.py: 2 launches of 8192 NAND gates, default parameters, one MI355X",
       "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/prof_target.py (one run per pass, tools/pmc_passes.sh)",
       "kernels": res}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
