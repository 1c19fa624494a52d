#!/bin/bash
mkdir -p gpurun_out/s35
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s35/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s35/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s35/pytest.log | head -10
REPS=3 STEPS=2 bash scripts/ab_bench.sh room 2>&1 | tail -4
