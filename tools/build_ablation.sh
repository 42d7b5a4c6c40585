#!/bin/bash
# Development aid: build variants of the library with -DSDPCUT_ABL_<X> switches of score.hip
# (results are WRONG by design; only the kernel time is of interest) into sdpcutsel_via_nn_amd/_abl/.
# usage: [ABL_SRC=eig] tools/build_ablation.sh NAME1[:FLAG,FLAG] NAME2 ...   e.g.  base nobias:NOBIAS notansig:NOTANSIG
# (ABL_SRC: the translation unit the switches apply to, default score)
set -e
cd "$(dirname "$0")/.."
P=sdpcutsel_via_nn_amd
SRC=${ABL_SRC:-score}
mkdir -p $P/_abl
python -m $P.build >/dev/null 2>&1
for spec in "$@"; do
  name=${spec%%:*}
  flags=""
  # FLAG -> -DSDPCUT_ABL_FLAG ; NAME=VALUE -> -DSDPCUT_NAME=VALUE (e.g. RING_DEPTH=6)
  if [[ "$spec" == *:* ]]; then for f in $(echo ${spec#*:} | tr , ' '); do
    if [[ "$f" == *=* ]]; then flags="$flags -DSDPCUT_$f"; else flags="$flags -DSDPCUT_ABL_$f"; fi; done; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c $P/csrc/$SRC.hip -o $P/_abl/${SRC}_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  objs=$(ls $P/csrc/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_abl/lib_$name.so $P/_abl/${SRC}_$name.o $objs
  echo built $P/_abl/lib_$name.so
done
