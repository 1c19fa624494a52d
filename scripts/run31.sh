#!/bin/bash
mkdir -p gpurun_out/s31
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s31/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s31/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s31/pytest.log | head -10
for rep in 1 2; do
for q in 0 1; do
  for sc in mis-spheres instances-10k; do
    RTX_GEN_MASKS=$q timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s31/g${q}_${sc}_$rep.json 2> gpurun_out/s31/g${q}_${sc}_$rep.err
    python scripts/ab_line.py "masks$q" $sc gpurun_out/s31/g${q}_${sc}_$rep.json
  done
done
done
