# usage: bash tools/ab_group_knobs.sh -> gpurun_out/ab_group_knobs.log: the two-group order loop at the headline size under its knobs, two rounds
mkdir -p gpurun_out
out=gpurun_out/ab_group_knobs.log; : > $out
for rep in 1 2; do
  for cfg in "" "SOSRT_SCAN_COLS=128" "SOSRT_SCAN_COLS=160" "SOSRT_SCAN_COLS=256" "SOSRT_GEMM_TAIL=128" "SOSRT_GEMM_TAIL=256" "SOSRT_GEMM_PAD_LDS=0 SOSRT_GROUP_RING_SLOTS=3" "SOSRT_GEMM_PAD_LDS=0 SOSRT_GROUP_RING_SLOTS=4"; do
    echo "cfg [$cfg]" >> $out
    env $cfg timeout -k 10 200 python bench.py --groups 2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || exit 1
  done
done
