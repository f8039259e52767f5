#!/usr/bin/env python3
"""Prints the sgm_* rows of rocprofv3 kernel-stats CSVs under a directory: calls and average microseconds."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgm_" in r["Name"]:
            n = r["Name"].replace("void stm::(anonymous namespace)::", "").replace("stm::(anonymous namespace)::", "").split("(")[0]
            print("  %-40s calls %4s avg_us %10.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
