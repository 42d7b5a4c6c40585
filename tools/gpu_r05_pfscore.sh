set -o pipefail
export TMPDIR=/tmp
# score-kernel variants of the fine histogram on one box: r4 tree, HEAD, HEAD with the adds gated by the launch (pf2), HEAD without (pf0)
for i in 1 2; do
  (cd _r4 && timeout -k 10 200 python bench.py --config c3 --steps 60) > gpurun_out/pfs_c3_r4_$i.json 2> gpurun_out/pfs.err || exit 1
  for v in head pf2 pf0; do
    lib=""; [ $v != head ] && lib=$PWD/sdpcutsel_via_nn_amd/_abl/lib_$v.so
    SDPCUT_LIB=$lib timeout -k 10 200 python bench.py --config c3 --steps 60 > gpurun_out/pfs_c3_${v}_$i.json 2>> gpurun_out/pfs.err || exit 1
    SDPCUT_LIB=$lib timeout -k 10 200 python bench.py --config c3 --steps 60 --no-prefilter > gpurun_out/pfs_c3_${v}off_$i.json 2>> gpurun_out/pfs.err || exit 1
  done
done
echo done
