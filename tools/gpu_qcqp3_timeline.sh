#!/bin/bash
# kernel timeline of a QCQP dim-3 round: tools/gpu_qcqp3_timeline.sh <qcqp golden npz> <round> <out name>   (SDPCUT_LIB=... for a variant library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$3 -o prof -f csv -- python3 tools/qcqp3_time.py $1 $2 40 > gpurun_out/prof_$3.log 2>&1 || exit 1
d=$(dirname $(find gpurun_out/prof_$3 -name prof_kernel_trace.csv | head -1))
python3 - "$d" > gpurun_out/$3.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + '/prof_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = [i for i, r in enumerate(rows) if 'point_copy' in r['Kernel_Name']]
def show(lo, hi, title):
    print(title)
    t0 = int(rows[lo]['Start_Timestamp'])
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print("%9.1f us  dur %7.1f  end %7.1f  queue %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:64]))
h = len(per) // 2
show(per[h - 5], per[h - 1], "--- rounds with the pairing on")
PY
cat gpurun_out/$3.txt
