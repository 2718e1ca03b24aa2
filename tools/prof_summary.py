#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals per step and the last step's BWT rounds."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('cjs::', '').replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t = tot.setdefault(n, [0, 0.0, 0.0])
    t[0] += 1; t[1] += d; t[2] = max(t[2], d)
allt = sum(v[1] for v in tot.values())
print("total kernel time per step: %.2f ms" % (allt / steps / 1e3))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-28s calls/step %6.1f  ms/step %8.3f  avg %9.1f us  max %9.1f us" % (n[:28], v[0] / steps, v[1] / steps / 1e3, v[1] / v[0], v[2]))
