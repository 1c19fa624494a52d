#!/bin/bash
mkdir -p gpurun_out/s20
RTX_CHUNK_SCATTER=1 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s20/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s20/pytest.log | cut -c1-200
for rep in 1 2; do
for q in 0 1; do
  for sc in blob mis room instances-10k; do
    RTX_CHUNK_SCATTER=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s20/c${q}_${sc}_$rep.json 2> gpurun_out/s20/c${q}_${sc}_$rep.err
    python scripts/ab_line.py "scatter$q" $sc gpurun_out/s20/c${q}_${sc}_$rep.json
  done
done
done
RTX_CHUNK_SCATTER=1 timeout 600 python scripts/exp_sort.py blob room 2>&1 | grep -v amdgpu.ids | head -8
