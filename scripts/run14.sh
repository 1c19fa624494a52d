#!/bin/bash
mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s14/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s14/pytest.log | cut -c1-200
REPS=3 STEPS=3 timeout 900 bash scripts/ab_bench.sh cornell > gpurun_out/s14/ab_cornell.log 2>&1; tail -2 gpurun_out/s14/ab_cornell.log
REPS=2 STEPS=3 timeout 900 bash scripts/ab_bench.sh blob > gpurun_out/s14/ab_blob.log 2>&1; tail -2 gpurun_out/s14/ab_blob.log
