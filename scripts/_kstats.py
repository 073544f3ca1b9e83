#!/usr/bin/env python3
"""Print the first rows of a rocprofv3 kernel_stats.csv: name, calls, average / min / max microseconds."""
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 4]:
    print(f'{r["Name"][:70]:70} calls {r["Calls"]:>6} avg {float(r["AverageNs"]) / 1e3:8.2f} us min {float(r["MinNs"]) / 1e3:8.2f} max {float(r["MaxNs"]) / 1e3:8.2f}')
