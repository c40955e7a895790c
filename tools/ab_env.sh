# usage: bash tools/ab_env.sh "ENV=VAL [ENV=VAL ...]" [groups] -> gpurun_out/ab_env.log: bench.py at several batch shapes without and with the settings, alternating
mkdir -p gpurun_out
out=gpurun_out/ab_env.log; : > $out
SET=$1; G=${2:-1}
run() {  # columns layers angles aerosol steps
  for rep in 1 2; do for cfg in "" "$SET"; do
    echo "cfg $1 $2 $3 $4 [$cfg]" >> $out
    env $cfg timeout -k 10 200 python bench.py --groups $G --columns $1 --layers $2 --angles $3 --aerosol $4 --steps $5 --warmup 2 --no-extras --no-cpu-baseline --pipelined 0 --check-columns 1 >> $out 2>&1 || return 1
  done; done
}
if [ -n "$SHAPES" ]; then eval "$SHAPES"; else run 512 200 128 eva 20 && run 256 200 128 eva 20 && run 1024 200 128 eva 10 && run 4096 200 128 eva 4 && run 512 200 256 eva 10 && run 4096 400 256 wildfire 3 && run 512 200 64 eva 20; fi
