#!/usr/bin/env python3
"""Print rocprofv3 kernel stats and the kernel timeline of the last bench step.
Usage: tools/timeline.py gpurun_out/prof_<tag>"""
import csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(d + '/prof_kernel_stats.csv')))
for r in rows[:14]:
    print("%-62s calls=%5s avg=%8.1f us pct=%s" % (r['Name'][:62], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
rows = list(csv.DictReader(open(d + '/prof_kernel_trace.csv')))
import os
if os.path.exists(d + '/prof_memory_copy_trace.csv'):      # --memory-copy-trace: SDMA transfers join the timeline
    for r in csv.DictReader(open(d + '/prof_memory_copy_trace.csv')):
        rows.append({'Start_Timestamp': r['Start_Timestamp'], 'End_Timestamp': r['End_Timestamp'],
                     'Kernel_Name': 'memcpy %s %s B' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '')))})
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'score_' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp']); prev = t0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us gap %6.1f dur %6.1f %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:44])); prev = e
