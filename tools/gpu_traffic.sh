#!/bin/bash
# HBM-side traffic of the scoring kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots).
set -o pipefail
export TMPDIR=/tmp
tag=${1:-traffic}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${tag}_$c -o pmc -- python3 tools/ablate.py 3 1000000 > gpurun_out/${tag}_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_$c.log; exit 1; }
done
python3 - <<'PY'
import csv, collections, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/*_%s/pmc_counter_collection.csv" % c)[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "score_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(c, k, "eig %.1f  nn %.1f  both %.1f  (counter units, mean of dispatches)" % (sum(v[2:8]) / 6, sum(v[10:16]) / 6, sum(v[18:24]) / 6))
PY
