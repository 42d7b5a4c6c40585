#!/bin/bash
# PMC counters of the scoring kernels (one rocprofv3 run per counter group; --pmc only with --kernel-trace).
set -o pipefail
export TMPDIR=/tmp
tag=${1:-pmc}
k=${2:-3}
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" \
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_IFETCH" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/${tag}_$i -o pmc -- python3 tools/ablate.py $k 200000 > gpurun_out/${tag}_$i.log 2>&1 || { tail -5 gpurun_out/${tag}_$i.log; exit 1; }
done
ls gpurun_out/${tag}_*/
