# usage: bash tools/ab_libs.sh "<lib> <lib> ..." [bench args...] -> gpurun_out/ab_libs.log: bench.py with each build (paths under
# sos-radiative-transfer_amd/), alternating, three rounds; prints ms per step and the contraction's ms per step of every run
mkdir -p gpurun_out
libs=$1; shift
out=gpurun_out/ab_libs.log; : > $out
for rep in 1 2 3; do
  for l in $libs; do
    echo "lib $l" >> $out
    SOSRT_LIB=$PWD/sos-radiative-transfer_amd/$l timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json
lib = None
for line in open("gpurun_out/ab_libs.log"):
    if line.startswith("lib "):
        lib = line.split()[1]
    elif line.startswith("{"):
        d = json.loads(line)
        k = d["kernel_ms_per_step"]
        print("%-28s %.3f ms per step   contraction %.3f   transport %.3f   check %s" % (lib, d["ms_per_step"], k["k_jn_gemm"], k["k_transport"], d["check"]["ok"]))
PY
