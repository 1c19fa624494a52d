#!/bin/bash
mkdir -p gpurun_out/s26
for rep in 1 2; do
for q in 28 26 24 22; do
  for sc in room mis; do
    RTX_PASS_LOG2=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s26/p${q}_${sc}_$rep.json 2> gpurun_out/s26/p${q}_${sc}_$rep.err
    python scripts/ab_line.py "pass$q" $sc gpurun_out/s26/p${q}_${sc}_$rep.json
  done
done
done
