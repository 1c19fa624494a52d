#!/bin/bash
mkdir -p gpurun_out/s44
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s44/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s44/pytest.log | cut -c1-300
grep -n "^E " gpurun_out/s44/pytest.log | head -10
REPS=4 STEPS=3 bash scripts/ab_bench.sh cornell 2>&1 | tail -3
REPS=2 STEPS=2 bash scripts/ab_bench.sh blob room 2>&1 | tail -5
