"""Condense the rocprofv3 --pmc dumps written by tools/pmc_passes.sh into one JSON per round.

  python3 tools/pmc_summarize.py <tag> [name]   reads gpurun_out/pmc_<tag>*/, writes profiles/<name>_pmc_summary.json
                                                (name defaults to tag); refuses to write a summary without the
                                                default workload's kernels

Per-launch averages of every counter for every kernel of the library that ran (kernel names shortened to
the function name); the default workload's kernels at the top level under "kernels", the other
workloads of tools/prof_target.py under "other_workloads"."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag


def short(name):
    m = re.search(r"(?:tfhe_amd::)?([A-Za-z0-9_]+_kernel)\b", name)
    return m.group(1) if m else None


def summarize(src):
    res = collections.defaultdict(dict)
    for f in sorted(glob.glob(f"{src}/*/*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            key = short(r["Kernel_Name"])
            if key:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            for c, v in d.items():
                res[k][c] = sum(v) / len(v)  # average per launch
                res[k]["launches"] = len(v)
    for k, d in res.items():
        if "FETCH_SIZE" in d:
            # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
            # wide coalesced stream (16 B per lane, which is how these kernels read their keys), so it
            # is doubled before it is compared with a byte count
            d["hbm_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2 + d.get("WRITE_SIZE", 0.0) * 1024
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d:
            d["valu_active_share_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in d:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1.0)
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    return res


out = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/prof_target.py <mode> "
                  "(one run per pass, tools/pmc_passes.sh <tag> <mode>)",
       "workload": "tools/prof_target.py default: 2 launches of 8192 NAND gates, default parameters, one MI355X",
       "kernels": summarize(f"gpurun_out/pmc_{tag}"), "other_workloads": {}}
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*")):
    if os.path.isdir(d):
        out["other_workloads"][d.split(f"pmc_{tag}_", 1)[1]] = summarize(d)
if not out["kernels"]:
    sys.exit(f"no counter dumps under gpurun_out/pmc_{tag}: nothing written")
json.dump(out, open(f"profiles/{name}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: sorted(v) for k, v in [("kernels", out["kernels"])] + list(out["other_workloads"].items())}, indent=1))
