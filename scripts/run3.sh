#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest.log 2>&1; tail -5 gpurun_out/s3/pytest.log
REPS=3 STEPS=3 bash scripts/ab_bench.sh cornell > gpurun_out/s3/ab_cornell.log 2>&1; tail -5 gpurun_out/s3/ab_cornell.log
for B in 64 128; do
  RTX_SHADE_BLOCK=$B python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s3/block$B.json 2>/dev/null
  python scripts/ab_line.py block$B cornell gpurun_out/s3/block$B.json
done
python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s3/block256.json 2>/dev/null
python scripts/ab_line.py block256 cornell gpurun_out/s3/block256.json
RTX_SHADE_BLOCK=64 python bench.py --scene room --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s3/room_block64.json 2>/dev/null
python scripts/ab_line.py block64 room gpurun_out/s3/room_block64.json
RTX_SHADE_BLOCK=128 python bench.py --scene room --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s3/room_block128.json 2>/dev/null
python scripts/ab_line.py block128 room gpurun_out/s3/room_block128.json
python bench.py --gpus 2 --devices 0,0 --scene cornell --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s3/multi2.json 2> gpurun_out/s3/multi2.err; tail -c 1500 gpurun_out/s3/multi2.json; tail -3 gpurun_out/s3/multi2.err
