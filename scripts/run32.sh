#!/bin/bash
mkdir -p gpurun_out/s32
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s32/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s32/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s32/pytest.log | head -10
for rep in 1 2 3; do
  for sc in mis-spheres instances-10k; do
    timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s32/g_${sc}_$rep.json 2> gpurun_out/s32/g_${sc}_$rep.err
    python scripts/ab_line.py "w4" $sc gpurun_out/s32/g_${sc}_$rep.json
  done
done
python scripts/exp_instances.py 100 3 16 2>&1 | tail -4
