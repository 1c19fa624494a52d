#!/bin/bash
mkdir -p gpurun_out/s29
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s29/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/s29/pytest.log | cut -c1-300
grep -n "Error\|assert" gpurun_out/s29/pytest.log | head -20
