#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output): mean KB per launch per tlfea
kernel.  usage: tools/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [out.csv]
(default output: profiles/r01_configB_pmc_hbm.csv, the round-1 file name)"""
import collections
import csv
import os
import sys

out = []
args = [a for a in sys.argv[1:] if a.endswith("counter_collection.csv")]
dst_arg = [a for a in sys.argv[1:] if not a.endswith("counter_collection.csv")]
for f in args:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "tlfea" in name:
            # the polynomial step kernel runs on two levels of the p-multigrid cycle: keep launches of different grid
            # sizes apart (fine level = the larger grid)
            if "cheb32_kernel" in name and r.get("Grid_Size"):
                name += " [grid %s]" % r["Grid_Size"]
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        out.append(dict(kernel=k, counter=c, launches=len(v), mean_KB=round(sum(v) / len(v), 2)))
dst = dst_arg[0] if dst_arg else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r01_configB_pmc_hbm.csv")
with open(dst, "w") as fh:
    w = csv.DictWriter(fh, fieldnames=["kernel", "counter", "launches", "mean_KB"])
    w.writeheader()
    w.writerows(out)
print(open(dst).read())
