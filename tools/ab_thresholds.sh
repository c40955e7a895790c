# usage: bash tools/ab_thresholds.sh [groups] -> gpurun_out/ab_thresholds.log: the live-count thresholds of the order loop at the headline size, three rounds
mkdir -p gpurun_out
G=${1:-1}
out=gpurun_out/ab_thresholds.log; : > $out
for rep in 1 2 3; do
  for cfg in "" "SOSRT_SCAN_COLS=256" "SOSRT_GEMM_SMALL=320" "SOSRT_SCAN_COLS=256 SOSRT_GEMM_SMALL=320" "SOSRT_GEMM_TAIL_FRAC=1.0"; do
    echo "cfg [$cfg]" >> $out
    env $cfg timeout -k 10 200 python bench.py --groups $G --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
