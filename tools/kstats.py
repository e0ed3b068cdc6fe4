#!/usr/bin/env python3
"""Prints calls / average ms of the tgx kernels in a rocprofv3 *_kernel_stats.csv (first match under a directory)."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not files:
        print(d, ": no kernel_stats.csv")
        continue
    print(d)
    for row in csv.DictReader(open(files[0])):
        n = row["Name"]
        if "tgx::" in n:
            print("   %-60s calls %5s  avg %8.3f ms" % (n.split("tgx::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e6))
