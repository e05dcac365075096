#!/usr/bin/env python3
"""print the top rows of a rocprofv3 *_kernel_stats.csv: tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f} pct {r['Percentage']}")
