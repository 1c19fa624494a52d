#!/bin/bash
mkdir -p gpurun_out/s15
for knob in "RTX_REFILL_MIN=16" "RTX_REFILL_MIN=8" "RTX_REFILL_MIN=32" "RTX_REFILL_MIN=48" "RTX_REFILL_MIN=64"; do
    env $knob timeout 300 python bench.py --scene instances-10k --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s15/inst_$knob.json 2> gpurun_out/s15/inst_$knob.err
    python scripts/ab_line.py "$knob" instances-10k gpurun_out/s15/inst_$knob.json
done
