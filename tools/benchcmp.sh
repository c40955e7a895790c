#!/bin/bash
# usage: benchcmp.sh tag "ENV=.. ENV=.." ...   (pairs)
while [ $# -gt 1 ]; do
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --check-columns 1 --pipelined 0 > gpurun_out/r2_b_$tag.log 2>&1
  python - "$tag" <<PY
import json,sys
tag=sys.argv[1]; f="gpurun_out/r2_b_%s.log"%tag
try:
    d=[json.loads(l) for l in open(f) if l.startswith("{")][0]
    k=d["kernel_ms_per_step"]
    print("%-22s %7.0f col/s %6.3f ms  gemm %.2f tr %.2f fo %.2f  ok=%s"%(tag, d["value"], d["ms_per_step"], k["k_jn_gemm"], k["k_transport"], k["k_first_order"], d["check"]["ok"]))
except Exception as e: print(tag, "ERR", e, open(f).read()[-800:])
PY
done
