#!/bin/bash
# short lists: the selection in one launch (tk_one_kernel, r4) against the multi-launch path (SDPCUT_ONE_KERNEL_SELECT=0)
for v in 1 0; do
  echo "== SDPCUT_ONE_KERNEL_SELECT=$v"
  for f in q_50_25_75_1 q_40_8_25_1 q_30_6_50_1 q_20_20_100_2; do
    SDPCUT_ONE_KERNEL_SELECT=$v python3 tools/qcqp3_time.py tests/golden/qcqp_rounds_${f}_s4.npz 2 300 2>/dev/null
  done
  SDPCUT_ONE_KERNEL_SELECT=$v python3 tools/round_time.py tests/golden/rounds_spar070_050_1_d5_s4.npz 2 2>/dev/null | tail -1
  SDPCUT_ONE_KERNEL_SELECT=$v python3 tools/round_time.py tests/golden/rounds_spar070_050_1_d5_s4.npz 9 2>/dev/null | tail -1
done
