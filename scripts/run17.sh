#!/bin/bash
mkdir -p gpurun_out/s17
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s17/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s17/pytest.log | cut -c1-200
for rep in 1 2; do
for q in 64 4 2; do
  for sc in room mis; do
    RTX_GUIDE_QUARTERS=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s17/g${q}_${sc}_$rep.json 2> gpurun_out/s17/g${q}_${sc}_$rep.err
    python scripts/ab_line.py "guide$q" $sc gpurun_out/s17/g${q}_${sc}_$rep.json
  done
done
done
