#!/bin/bash
# Several settings of environment knobs on ONE box, interleaved: bash scripts/ab_envs.sh <scene> <reps> "A=1" "A=2 B=3" ...   ("-" = the default environment)
SC=$1; REPS=$2; shift 2
mkdir -p gpurun_out/ab
for rep in $(seq 1 $REPS); do
  i=0
  for K in "$@"; do
    i=$((i+1))
    if [ "$K" = "-" ]; then E=""; else E="$K"; fi
    env $E python bench.py --scene $SC --steps 2 --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/env${i}_${SC}_$rep.json > /dev/null 2> gpurun_out/ab/env${i}_${SC}_$rep.err
    python scripts/ab_line.py "$K" $SC gpurun_out/ab/env${i}_${SC}_$rep.json
  done
done
