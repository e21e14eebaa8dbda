#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --kernel-trace --stats `*kernel_stats.csv` (microseconds)."""
import csv, glob, sys
pat = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for f in sorted(glob.glob(pat, recursive=True)):
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:top]:
        print(f"{r['Name'][:72]:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} min {float(r['MinNs']) / 1e3:9.1f} "
              f"max {float(r['MaxNs']) / 1e3:9.1f} total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):5.1f}%")
