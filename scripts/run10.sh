#!/bin/bash
mkdir -p gpurun_out/s10
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest.log 2>&1; tail -3 gpurun_out/s10/pytest.log | cut -c1-200
REPS=2 STEPS=2 timeout 1200 bash scripts/ab_bench.sh room > gpurun_out/s10/ab_room.log 2>&1; tail -3 gpurun_out/s10/ab_room.log
REPS=2 STEPS=3 timeout 600 bash scripts/ab_bench.sh mis > gpurun_out/s10/ab_mis.log 2>&1; tail -3 gpurun_out/s10/ab_mis.log
