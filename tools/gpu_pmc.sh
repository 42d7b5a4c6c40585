#!/bin/bash
# PMC counters of the scoring kernel (one rocprofv3 run per counter group; --pmc only with --kernel-trace).
# usage: tools/gpu_pmc.sh <tag> [k] [count] [variant] [flagset] [kernel name pattern]
set -o pipefail
export TMPDIR=/tmp
tag=${1:-pmc}
k=${2:-3}
count=${3:-1000000}
variant=${4:-mfma}
flagset=${5:-eig+nn}
pat=${6:-score_}
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" \
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/${tag}_$i -o pmc -- python3 tools/ablate.py $k $count 100 $variant $flagset > gpurun_out/${tag}_$i.log 2>&1 || { tail -5 gpurun_out/${tag}_$i.log; exit 1; }
  tail -1 gpurun_out/${tag}_$i.log
done
python3 tools/pmc_summary.py $pat gpurun_out/${tag}_1 gpurun_out/${tag}_2 gpurun_out/${tag}_3 | tee gpurun_out/${tag}_summary.txt
