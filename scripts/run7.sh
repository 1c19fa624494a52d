#!/bin/bash
mkdir -p gpurun_out/s7
python -m pytest tests -m gpu -x -q > gpurun_out/s7/pytest.log 2>&1; tail -3 gpurun_out/s7/pytest.log | cut -c1-300
REPS=2 STEPS=3 bash scripts/ab_bench.sh mis > gpurun_out/s7/ab_mis.log 2>&1; tail -4 gpurun_out/s7/ab_mis.log
REPS=1 STEPS=2 bash scripts/ab_bench.sh room > gpurun_out/s7/ab_room.log 2>&1; tail -4 gpurun_out/s7/ab_room.log
for knob in "RTX_LEAF_MIN=32" "RTX_LEAF_MIN=48" "RTX_LEAF_MIN=64"; do
    env $knob python bench.py --scene instances-10k --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s7/inst_$knob.json 2> gpurun_out/s7/inst_$knob.err
    python scripts/ab_line.py "$knob" instances-10k gpurun_out/s7/inst_$knob.json
done
