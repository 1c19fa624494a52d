#!/bin/bash
mkdir -p gpurun_out/s22
for rep in 1 2; do
for q in 0 14 10; do
  for sc in blob mis room; do
    RTX_TRACE_WAVES_PER_CU=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s22/w${q}_${sc}_$rep.json 2> gpurun_out/s22/w${q}_${sc}_$rep.err
    python scripts/ab_line.py "waves$q" $sc gpurun_out/s22/w${q}_${sc}_$rep.json
  done
done
for sc in blob mis room; do
    RTX_TOP_BLOCKS_PER_CU=2 timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s22/t2_${sc}_$rep.json 2> gpurun_out/s22/t2_${sc}_$rep.err
    python scripts/ab_line.py "top2" $sc gpurun_out/s22/t2_${sc}_$rep.json
done
done
