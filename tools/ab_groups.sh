# usage: bash tools/ab_groups.sh  -> gpurun_out/ab_groups.log: bench.py at several batch shapes with one and two column groups, alternating
mkdir -p gpurun_out
out=gpurun_out/ab_groups.log; : > $out
run() {  # columns layers angles aerosol steps
  for rep in 1 2; do for gq in 1 2; do
    echo "cfg $1 $2 $3 $4 groups=$gq" >> $out
    timeout -k 10 200 python bench.py --groups $gq --columns $1 --layers $2 --angles $3 --aerosol $4 --steps $5 --warmup 2 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || return 1
  done; done
}
if [ -n "$SHAPES" ]; then eval "$SHAPES"; else
run 256 200 128 eva 20 && run 384 200 128 eva 20 && run 1024 200 128 eva 10 && run 4096 200 128 eva 4 && run 512 200 256 eva 10 && run 512 400 256 wildfire 6 && run 4096 400 256 wildfire 3 && run 512 200 64 eva 20 && run 512 200 128 hg 20
fi
