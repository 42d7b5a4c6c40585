#!/bin/bash
# the QCQP round of bench.py's secondary.c5 (q_50_10_25_1, dim 5, two device-built covers): numbers + kernel timeline
# (two handles = two streams: with the pairing on, the constraints cover's kernels start while the objective cover's run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/c5_round.py 200 > gpurun_out/r03_c5_round.json 2> gpurun_out/r03_c5_round.err || exit 1
rm -rf gpurun_out/prof_r03_c5b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_c5b -o prof -f csv -- python3 tools/c5_round.py 30 > gpurun_out/prof_r03_c5b.log 2>&1 || exit 1
d=$(dirname $(find gpurun_out/prof_r03_c5b -name prof_kernel_trace.csv | head -1))
python3 - "$d" > gpurun_out/r03_c5_timeline.txt <<'PY'
import csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(d + '/prof_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the overlapped strategy-4 rounds are the first timed block: take three rounds from the middle of it, then three of the last block
def show(lo, hi, title):
    print(title)
    t0 = int(rows[lo]['Start_Timestamp'])
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print("%9.1f us  dur %7.1f  queue %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
n = len(rows)
per = [i for i, r in enumerate(rows) if 'point_copy' in r['Kernel_Name']]
print("%d kernels, %d rounds (point copies)" % (n, len(per)))
q = len(per) // 8
show(per[q], per[q + 4], "--- two QCQP rounds, pairing ON (strategy 4 block): kernels of both handles by start time")
show(per[-5], per[-1], "--- two QCQP rounds, pairing OFF (last block, strategy 1)")
PY
head -c 3000 gpurun_out/r03_c5_round.json; echo; head -n 60 gpurun_out/r03_c5_timeline.txt
