#!/bin/bash
mkdir -p gpurun_out/s9
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/s9/pytest.log 2>&1; tail -20 gpurun_out/s9/pytest.log | cut -c1-200
