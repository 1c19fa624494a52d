#!/bin/bash
mkdir -p gpurun_out/s48
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s48/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s48/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s48/pytest.log | head -10
for rep in 1 2 3; do
for q in 0 1; do
    RTX_SHADE_LEAN=$q timeout 300 python bench.py --scene mis --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s48/l${q}_$rep.json 2> gpurun_out/s48/l${q}_$rep.err
    python scripts/ab_line.py "lean$q" mis gpurun_out/s48/l${q}_$rep.json
done
done
