# usage: bash tools/trace_orders.sh <tag> [ENV=VAL ...]   -> gpurun_out/order_table_<tag>.txt (durations of the kernels of one solve)
tag=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 tools/run_once.py 512 3 > gpurun_out/trace_$tag.log 2>&1
f=$(find gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/order_table.py $f 1 > gpurun_out/order_table_$tag.txt
rm -rf gpurun_out/trace_$tag
python3 - $tag <<'PY'
import sys
rows=[l.split() for l in open('gpurun_out/order_table_%s.txt'%sys.argv[1]).read().splitlines()[2:]]
g=[];t=[]
for r in rows:
    if r[1].startswith('k_') or r[1].startswith('__'): continue
    (g if len(g)==len(t) else t).append(float(r[2]))
print(sys.argv[1], "gemm", " ".join("%.0f"%x for x in g)); print(sys.argv[1], "tr  ", " ".join("%.0f"%x for x in t)); print("sum gemm %.0f tr %.0f"%(sum(g),sum(t)))
PY
