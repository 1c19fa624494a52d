#!/bin/bash
mkdir -p gpurun_out/s12
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s12/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s12/pytest.log | cut -c1-200
for sc in cornell blob mis room; do
  timeout 300 python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s12/$sc.json 2> gpurun_out/s12/$sc.err
  python scripts/ab_line.py skipdead $sc gpurun_out/s12/$sc.json
  RTX_CAST_DEAD_RAYS=1 timeout 300 python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s12/${sc}_cast.json 2> gpurun_out/s12/${sc}_cast.err
  python scripts/ab_line.py castdead $sc gpurun_out/s12/${sc}_cast.json
done
