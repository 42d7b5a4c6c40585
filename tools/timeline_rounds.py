#!/usr/bin/env python3
"""Kernel timeline of the last rounds of a rocprofv3 kernel trace whose rounds start with point_copy_kernel
(every fused round does).  Usage: tools/timeline_rounds.py <dir with prof_kernel_trace.csv> [rounds]"""
import csv
import sys
d = sys.argv[1]
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(d + '/prof_kernel_stats.csv')))
for r in rows[:12]:
    print("%-70s calls=%5s avg=%8.1f us pct=%s" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
rows = list(csv.DictReader(open(d + '/prof_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'point_copy' in r['Kernel_Name']]
a = starts[-nr - 1]
b = starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'point_copy' in r['Kernel_Name']:
        print("---- round: %.1f us after the previous one's first kernel" % ((s - t0) / 1e3))
        t0 = s
    print("%8.1f us gap %6.1f dur %6.1f %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:60]))
    prev = e
