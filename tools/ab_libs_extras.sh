# usage: bash tools/ab_libs_extras.sh "<lib> <lib> ..." -> the headline and the extras of bench.py with each build, alternating, two rounds
mkdir -p gpurun_out
out=gpurun_out/ab_libs_extras.log; : > $out
for rep in 1 2; do
  for l in $1; do
    echo "lib $l" >> $out
    SOSRT_LIB=$PWD/sos-radiative-transfer_amd/$l timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json
lib = None
for line in open("gpurun_out/ab_libs_extras.log"):
    if line.startswith("lib "):
        lib = line.split()[1]
    elif line.startswith("{"):
        d = json.loads(line); e = d["extras"]
        print("%-22s step %.3f ms  two groups %.3f  c2 %.4f  c3 %.4f  c4_shard %.3f  shipped %.3f  c5 %.1f ms" % (
            lib, d["ms_per_step"], d["two_groups"]["ms_per_step"], e["c2"]["ms_per_solve"], e["c3"]["ms_per_solve"],
            e["c4_shard"]["ms_per_solve"], e["shipped"]["ms_per_solve"], e["c5"]["ms_per_solve"]))
PY
