#!/bin/bash
mkdir -p gpurun_out/s21
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s21/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s21/pytest.log | cut -c1-200
for rep in 1 2; do
for q in 32 0; do
  for sc in blob mis room; do
    RTX_STACK_VARIANT=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s21/v${q}_${sc}_$rep.json 2> gpurun_out/s21/v${q}_${sc}_$rep.err
    python scripts/ab_line.py "stack$q" $sc gpurun_out/s21/v${q}_${sc}_$rep.json
  done
done
done
