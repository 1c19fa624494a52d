#!/bin/bash
mkdir -p gpurun_out/s34
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s34/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s34/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s34/pytest.log | head -10
REPS=2 STEPS=2 bash scripts/ab_bench.sh cornell blob mis room 2>&1 | tail -14
