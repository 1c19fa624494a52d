#!/bin/bash
mkdir -p gpurun_out/s8
python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1; tail -3 gpurun_out/s8/pytest.log | cut -c1-300
REPS=3 STEPS=3 bash scripts/ab_bench.sh cornell > gpurun_out/s8/ab_cornell.log 2>&1; tail -4 gpurun_out/s8/ab_cornell.log
for B in 64 128; do
  RTX_SHADE_BLOCK=$B python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s8/block$B.json 2>/dev/null
  python scripts/ab_line.py block$B cornell gpurun_out/s8/block$B.json
done
REPS=1 STEPS=2 bash scripts/ab_bench.sh mis room > gpurun_out/s8/ab2.log 2>&1; tail -7 gpurun_out/s8/ab2.log
RTX_SHADE_BLOCK=128 python bench.py --scene room --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s8/room_block128.json 2>/dev/null
python scripts/ab_line.py block128 room gpurun_out/s8/room_block128.json
