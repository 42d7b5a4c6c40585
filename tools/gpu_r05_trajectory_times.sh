set -o pipefail
export TMPDIR=/tmp
# separation time of every recorded round of the reference trajectories: this tree and the round-4 tree (_r4/) alternating on ONE box
out=gpurun_out/r05_separation_time_per_recorded_round.txt
: > $out
for t in rounds_spar125_075_1_d4_s4 rounds_spar125_075_1_d3_s2 rounds_spar100_050_1_d5_s4 rounds_spar070_050_1_d5_s4 rounds_spar080_075_1_d4_s1; do
  printf "r5     : " >> $out; timeout -k 10 200 python3 tools/trajectory_times.py $t 15 2>>gpurun_out/r05_traj.err | tail -1 >> $out || exit 1
  printf "r4 tree: " >> $out; (cd _r4 && timeout -k 10 200 python3 tools/trajectory_times.py $t 15 2>>../gpurun_out/r05_traj.err | tail -1) >> $out || exit 1
done
cat $out
