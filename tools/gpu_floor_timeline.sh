#!/bin/bash
# kernel timeline of fused rounds on short synthetic lists: tools/gpu_floor_timeline.sh <list length> <out name>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$2
rocprofv3 --kernel-trace -d gpurun_out/prof_$2 -o prof -f csv -- python3 tools/round_floor.py 30 $1 > gpurun_out/prof_$2.log 2>&1 || exit 1
d=$(dirname $(find gpurun_out/prof_$2 -name prof_kernel_trace.csv | head -1))
python3 - "$d" > gpurun_out/$2.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + '/prof_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = [i for i, r in enumerate(rows) if 'point_copy' in r['Kernel_Name']]
def show(lo, hi, title):
    print(title)
    t0 = int(rows[lo]['Start_Timestamp'])
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print("%9.1f us  dur %7.1f  end %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r['Kernel_Name'][:70]))
n = len(per) // 3
show(per[n - 3], per[n - 1], "--- two combined rounds")
show(per[2 * n - 3], per[2 * n - 1], "--- two feasibility rounds")
show(per[-3], per[-1], "--- two optimality rounds")
PY
cat gpurun_out/$2.txt
