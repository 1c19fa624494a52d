#!/bin/bash
mkdir -p gpurun_out/s19
RTX_OCTANT_SHARDS=1 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s19/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s19/pytest.log | cut -c1-200
for rep in 1 2; do
for q in 0 1; do
  for sc in cornell blob mis room; do
    RTX_OCTANT_SHARDS=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s19/o${q}_${sc}_$rep.json 2> gpurun_out/s19/o${q}_${sc}_$rep.err
    python scripts/ab_line.py "octant$q" $sc gpurun_out/s19/o${q}_${sc}_$rep.json
  done
done
done
