"""Shrink the rocprofv3 counter_collection CSVs of a profile round in place: one row per (kernel, counter) with the mean
Counter_Value over the kernel's dispatches (the other columns from its first dispatch) -- what tools/summarize_profiles.py averages
anyway; the raw files (a row per dispatch and counter: tens of MB per round) do not fit the 64 MiB that travel back from the GPU box.
usage: python tools/reduce_counters.py gpurun_out/prof_r03"""
import csv
import sys
from pathlib import Path

for path in sorted(Path(sys.argv[1]).glob("*counter_collection.csv")):
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        fields = rd.fieldnames
        first, total, count = {}, {}, {}
        for r in rd:
            key = (r["Kernel_Name"], r["Counter_Name"])
            if key not in first:
                first[key] = r
                total[key] = 0.0
                count[key] = 0
            total[key] += float(r["Counter_Value"])
            count[key] += 1
    with open(path, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=fields + ["Dispatches"], quoting=csv.QUOTE_NONNUMERIC)
        wr.writeheader()
        for key, r in first.items():
            row = dict(r)
            row["Counter_Value"] = total[key] / count[key]
            row["Dispatches"] = count[key]
            wr.writerow(row)
