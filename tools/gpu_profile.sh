#!/bin/bash
# Profiles of one bench configuration on the GPU box (rocprofv3; counters in their own runs, with
# --kernel-trace only).  usage: tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/<tag>_*
#   1. kernel trace + stats of the bench command          -> <tag>_kernel_stats.csv, <tag>_step_timeline.txt, <tag>_bench.json
#   2. three PMC groups of the score kernel               -> <tag>_score_kernel_pmc.txt
#   3. FETCH_SIZE / WRITE_SIZE in separate passes         -> <tag>_score_kernel_hbm_traffic.txt
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out
mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/prof_$tag -o prof -- $B --steps 200 --warmup 10 "$@" > $out/${tag}_bench.json 2> $out/prof_$tag.err || { tail -20 $out/prof_$tag.err; exit 1; }
cp $out/prof_$tag/prof_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 tools/timeline.py $out/prof_$tag > $out/${tag}_step_timeline.txt
tail -12 $out/${tag}_step_timeline.txt
i=0
: > $out/${tag}_score_kernel_pmc.txt
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" \
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_${tag}_$i -o pmc -- $B --steps 30 --warmup 3 "$@" > $out/pmc_${tag}_$i.log 2>&1 || { tail -5 $out/pmc_${tag}_$i.log; exit 1; }
done
python3 tools/pmc_summary.py score_mfma $out/pmc_${tag}_1 $out/pmc_${tag}_2 $out/pmc_${tag}_3 | tee -a $out/${tag}_score_kernel_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/traffic_${tag}_$c -o pmc -- $B --steps 30 --warmup 3 "$@" > $out/traffic_${tag}_$c.log 2>&1 || { tail -5 $out/traffic_${tag}_$c.log; exit 1; }
done
python3 - "$tag" <<'PY' | tee $out/${tag}_score_kernel_hbm_traffic.txt
import csv, collections, glob, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/traffic_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[(k, c)] = sum(v) / len(v)
        print(c, k, "%.1f KiB per dispatch (mean of %d launches)" % (out[(k, c)], len(v)))
for k in sorted({k for k, _ in out}):
    f, w = out.get((k, "FETCH_SIZE"), 0.0), out.get((k, "WRITE_SIZE"), 0.0)
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section)
    print("%s: fetch 2 x %.1f KiB + write %.1f KiB = %.2f MB per launch" % (k, f, w, (2 * f + w) * 1024 / 1e6))
PY
