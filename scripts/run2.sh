#!/bin/bash
mkdir -p gpurun_out/s2
python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; tail -3 gpurun_out/s2/pytest.log
bash scripts/ab_bench.sh cornell blob mis room > gpurun_out/s2/ab.log 2>&1; cat gpurun_out/s2/ab.log
