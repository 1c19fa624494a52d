#!/bin/bash
mkdir -p gpurun_out/s30
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s30/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s30/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s30/pytest.log | head -10
for rep in 1 2; do
for q in 1 0; do
    RTX_MIS_REACH=$q timeout 300 python bench.py --scene mis-spheres --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s30/m${q}_$rep.json 2> gpurun_out/s30/m${q}_$rep.err
    python scripts/ab_line.py "reach$q" mis-spheres gpurun_out/s30/m${q}_$rep.json
done
done
