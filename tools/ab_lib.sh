# usage: bash tools/ab_lib.sh <other lib> [bench args...] -> gpurun_out/ab_lib.log: bench.py with libsosrt.so and with another build, alternating, three rounds
mkdir -p gpurun_out
other=$1; shift
out=gpurun_out/ab_lib.log; : > $out
for rep in 1 2 3; do
  for l in libsosrt.so $other; do
    echo "lib $l" >> $out
    SOSRT_LIB=$PWD/sos-radiative-transfer_amd/$l timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
