#!/bin/bash
mkdir -p gpurun_out/s49
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s49/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s49/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s49/pytest.log | head -10
for rep in 1 2 3; do
for q in 0 1; do
    RTX_SHADE_BOUNCED=$q timeout 300 python bench.py --scene room --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s49/b${q}_$rep.json 2> gpurun_out/s49/b${q}_$rep.err
    python scripts/ab_line.py "bounced$q" room gpurun_out/s49/b${q}_$rep.json
done
done
