set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
# 1. kernel stats of the bench command (python3 directly after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --pipelined 0 > $O/bench_under_rocprof.log 2>&1
echo "stats done"
# 2. PMC passes over one solve each (separate runs, kernel-trace only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_once.py 512 1 > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_once.py 512 1 > $O/pmc_write.log 2>&1
echo "write done"
# 3. kernel trace of plain solves for the order table
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/run_once.py 512 3 > $O/trace.log 2>&1
echo "trace done"
find $O -name "*.csv" | head -20
# 4. reductions (copied into profiles/ by hand afterwards)
python3 tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) 1 > $O/pmc_traffic.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep '^{' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
python3 tools/order_table.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 1 > $O/order_table.txt
# 4b. the order loop the library takes by itself at this size: two column groups on two streams (SOSRT_GROUPS=0 = auto) -- kernel
#     stats of the same bench command, HBM traffic of one solve, overlap of the two groups' kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 bench.py --groups auto --steps 5 --warmup 1 --no-cpu-baseline --no-extras --pipelined 0 > $O/bench_two_groups_under_rocprof.log 2>&1
cp $(find $O/stats2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_two_groups.csv
grep '^{' $O/bench_two_groups_under_rocprof.log > $O/bench_two_groups_under_rocprof.json
python3 tools/overlap.py $(find $O/stats2 -name "*kernel_trace.csv" | head -1) > $O/two_groups_overlap.txt
export SOSRT_GROUPS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch2 -- python3 tools/run_once.py 512 1 > $O/pmc_fetch2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write2 -- python3 tools/run_once.py 512 1 > $O/pmc_write2.log 2>&1
python3 tools/pmc_traffic.py $(find $O/pmc_fetch2 -name "*counter_collection.csv" | head -1) $(find $O/pmc_write2 -name "*counter_collection.csv" | head -1) 1 > $O/pmc_traffic_two_groups.json
unset SOSRT_GROUPS
rm -rf $O/stats2 $O/pmc_fetch2 $O/pmc_write2
echo "two groups done"
# 5. the plain bench line (default flags)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/trace
ls -la $O
