#!/bin/bash
# score kernel time over the list length, this build against -DSDPCUT_BALANCED_TAIL=0 (_abl/lib_nobal.so), same box:
# fills of the last round 0.25 .. 1.0 at one full round, 0.85 .. 0.97 at two and three
for n in 655360 838860 917504 956825 983040 996147 1000000 1022361 1048576 1494221 1530921 1557135 2055209 2081423; do
  for name in default nobal; do
    lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
    [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
    echo "$name: $(SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py 3 $n 100 mfma eig+nn 2>/dev/null)"
  done
done
for k in 2 4 5; do
  for name in default nobal; do
    lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
    [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
    echo "$name: $(SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig+nn 2>/dev/null)"
  done
done
