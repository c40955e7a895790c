# (needs the order loop of tools/cu_split_order_loop.patch: git apply it and rebuild -- the library itself has no SOSRT_CU_SPLIT)
# usage: bash tools/ab_cu_split.sh [bench args...] -> gpurun_out/ab_cu_split.log + digest: the two-group order loop with the kernels of the
# dense orders on disjoint sets of CUs (SOSRT_CU_SPLIT = CUs of the contraction's part), against the shared-CU default, two rounds
mkdir -p gpurun_out
out=gpurun_out/ab_cu_split.log; : > $out
for rep in 1 2; do
  for cfg in "" "SOSRT_CU_SPLIT=128" "SOSRT_CU_SPLIT=152" "SOSRT_CU_SPLIT=168" "SOSRT_CU_SPLIT=184" "SOSRT_CU_SPLIT=168 SOSRT_CU_SPLIT_FRAC=0.3" "SOSRT_CU_SPLIT=168 SOSRT_CU_SPLIT_FRAC=0.8"; do
    echo "cfg [$cfg]" >> $out
    env $cfg timeout -k 10 200 python bench.py --groups auto --steps 20 --warmup 3 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 "$@" >> $out 2>&1 || exit 1
  done
done
python3 tools/ab_env_digest.py $out
