#!/bin/bash
mkdir -p gpurun_out/s47
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s47/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s47/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s47/pytest.log | head -10
for rep in 1 2 3; do
    timeout 300 python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s47/c_$rep.json 2> gpurun_out/s47/c_$rep.err
    python scripts/ab_line.py "recip" cornell gpurun_out/s47/c_$rep.json
done
