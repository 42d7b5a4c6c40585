#!/bin/bash
# HBM-side traffic of the scoring kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots).
# usage: tools/gpu_traffic.sh <tag> [variant]      (variant: mfma | valu | simple; eig+nn launches only)
set -o pipefail
export TMPDIR=/tmp
tag=${1:-traffic}
variant=${2:-mfma}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${tag}_$c -o pmc -- python3 tools/ablate.py 3 1000000 100 $variant eig+nn > gpurun_out/${tag}_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_$c.log; exit 1; }
done
python3 - "$tag" <<'PY'
import csv, collections, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/%s_%s/pmc_counter_collection.csv" % (tag, c))):
        if "score_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[(k, c)] = sum(v) / len(v)
        print(c, k, "%.1f KiB per dispatch (mean of %d eig+nn launches)" % (out[(k, c)], len(v)))
for k in sorted({k for k, _ in out}):
    f, w = out.get((k, "FETCH_SIZE"), 0.0), out.get((k, "WRITE_SIZE"), 0.0)
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section)
    print("%s: fetch 2 x %.1f KiB + write %.1f KiB = %.2f MB per launch" % (k, f, w, (2 * f + w) * 1024 / 1e6))
PY
