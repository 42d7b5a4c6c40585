#!/bin/bash
# bench (c2, k = 2..5 secondaries) of library variants built by tools/build_ablation.sh: usage tools/gpu_abl_bench.sh name1 name2 ...
for name in "$@"; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  SDPCUT_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-c3 --steps 100 2>/dev/null > gpurun_out/abl_$name.json
  python3 - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abl_%s.json" % sys.argv[1]))
s = d["secondary"]
print("%-10s k3 step %.1f us kernel %.1f us frac %.3f | k2 %.1f (%.3f) k4 %.1f (%.3f) k5 %.1f (%.3f)" % (
    sys.argv[1], d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"],
    s["k2"]["kernel_ms"] * 1e3, s["k2"]["roofline_frac"], s["k4"]["kernel_ms"] * 1e3, s["k4"]["roofline_frac"],
    s["k5"]["kernel_ms"] * 1e3, s["k5"]["roofline_frac"]))
PY
done
