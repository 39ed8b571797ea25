#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/prof.sh into one text summary (per-kernel time + PMC sums)."""
import collections
import csv
import glob
import os
import sys


def main():
    out = sys.argv[1]
    for f in glob.glob(os.path.join(out, "kt", "**", "*_kernel_stats.csv"), recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats)")
        for r in csv.DictReader(open(f)):
            print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):14.0f} total_ns={r['TotalDurationNs']:>14s} pct={r['Percentage']}")
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            agg = collections.OrderedDict()
            meta = {}
            for r in rows:
                name = r["Kernel_Name"]
                want = os.environ.get("RRT_PROF_KERNEL", "k_render")  # kernel the summary is about
                if want not in name:
                    continue
                if want == "k_render":
                    # k_render<MODE, COUNT, PIPELINED>: only the plain (non-counting) form, which is the one that is timed
                    args = name.split("k_render")[1].replace(" ", "").strip("<").split(">")[0].split(",")
                    if len(args) >= 2 and args[1] not in ("false", "0"):
                        continue
                key = (name[:48], r["Dispatch_Id"])
                agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
                meta[key] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
            print(f"== {os.path.basename(d)}")
            for key, vals in agg.items():
                print(" ", key[0], "dispatch", key[1], "vgpr/agpr/sgpr/lds/grid/wg =", meta[key])
                for k, v in vals.items():
                    print(f"      {k:32s} {v:16.6g}")


if __name__ == "__main__":
    main()
