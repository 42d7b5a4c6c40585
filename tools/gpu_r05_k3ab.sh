set -o pipefail
export TMPDIR=/tmp
for i in 1 2 3; do
  (cd _r4 && timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-secondary --no-cpu-baseline) > gpurun_out/k3ab_r4_$i.json 2> gpurun_out/k3ab.err || exit 1
  for v in head pfni; do
    lib=""; [ $v != head ] && lib=$PWD/sdpcutsel_via_nn_amd/_abl/lib_$v.so
    SDPCUT_LIB=$lib timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/k3ab_${v}_$i.json 2>> gpurun_out/k3ab.err || exit 1
  done
done
echo done
