#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv compactly: python tools/kernel_stats.py <csv> [divide-by]"""
import csv
import sys

div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("aloha::k::(anonymous namespace)::", "").replace("aloha::k::", "")[:64]
    print(f"{name:66s} calls {int(r['Calls']) / div:7.1f}  total_us {int(r['TotalDurationNs']) / 1e3 / div:9.1f}  avg_us {float(r['AverageNs']) / 1e3:8.1f}")
    tot += int(r["TotalDurationNs"])
print(f"total kernel time: {tot / 1e3 / div:.1f} us" + (f" per 1/{div:g} of the run" if div != 1 else ""))
