# usage: bash tools/ab_env_values.sh VAR "v1 v2 ..." [bench args...]: bench.py with VAR set to each value, alternating, three rounds
var=$1; vals=$2; shift; shift
mkdir -p gpurun_out; out=gpurun_out/ab_env_values.log; : > $out
for rep in 1 2 3; do
  for v in $vals; do
    echo "val $v" >> $out
    env $var=$v timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json
v = None
for line in open("gpurun_out/ab_env_values.log"):
    if line.startswith("val "):
        v = line.split()[1]
    elif line.startswith("{"):
        d = json.loads(line); k = d["kernel_ms_per_step"]
        print("%-8s %.3f ms per step   contraction %.3f   transport %.3f" % (v, d["ms_per_step"], k["k_jn_gemm"], k["k_transport"]))
PY
