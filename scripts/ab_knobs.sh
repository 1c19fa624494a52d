#!/bin/bash
# A/B of several environment settings on ONE box, interleaved: bash scripts/ab_knobs.sh "scene ..." "A=1 B=2" "A=2" ... ("-" = the default); REPS rounds (default 2)
SCENES=$1; shift
REPS=${REPS:-2}
mkdir -p gpurun_out/ab
for rep in $(seq 1 $REPS); do
  n=0
  for KN in "$@"; do
    n=$((n+1)); name=v$n
    for sc in $SCENES; do
      if [ "$KN" = "-" ]; then
        python bench.py --scene $sc --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/${name}_${sc}_$rep.json > gpurun_out/ab/${name}_${sc}_$rep.line 2> gpurun_out/ab/${name}_${sc}_$rep.err
      else
        env $KN python bench.py --scene $sc --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/${name}_${sc}_$rep.json > gpurun_out/ab/${name}_${sc}_$rep.line 2> gpurun_out/ab/${name}_${sc}_$rep.err
      fi
      python scripts/ab_line.py "$KN" $sc gpurun_out/ab/${name}_${sc}_$rep.json
    done
  done
done
