#!/bin/bash
mkdir -p gpurun_out/s11
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest.log 2>&1; tail -3 gpurun_out/s11/pytest.log | cut -c1-200
REPS=3 STEPS=3 timeout 900 bash scripts/ab_bench.sh cornell > gpurun_out/s11/ab_cornell.log 2>&1; tail -2 gpurun_out/s11/ab_cornell.log
REPS=2 STEPS=3 timeout 900 bash scripts/ab_bench.sh blob > gpurun_out/s11/ab_blob.log 2>&1; tail -2 gpurun_out/s11/ab_blob.log
