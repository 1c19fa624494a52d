#!/bin/bash
mkdir -p gpurun_out/s16
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s16/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s16/pytest.log | cut -c1-200
for knob in "RTX_OBJ_PAIRS=1" "RTX_OBJ_PAIRS=0" "RTX_OBJ_PAIRS=1" "RTX_OBJ_PAIRS=0"; do
    env $knob timeout 300 python bench.py --scene instances-10k --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s16/inst_$knob.json 2> gpurun_out/s16/inst_$knob.err
    python scripts/ab_line.py "$knob" instances-10k gpurun_out/s16/inst_$knob.json
done
python scripts/exp_instances.py 100 3 16 2>&1 | tail -4
