# usage: bash tools/trace_one.sh <tag> <columns> <angles> [ENV=VAL ...]  -> gpurun_out/order_table_<tag>.txt (kernels of one solve, with gaps)
tag=$1; cols=$2; ang=$3; shift 3
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 tools/run_once.py $cols 3 $ang > gpurun_out/trace_$tag.log 2>&1
f=$(find gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/order_table.py $f 1 > gpurun_out/order_table_$tag.txt
rm -rf gpurun_out/trace_$tag
