#!/bin/bash
mkdir -p gpurun_out/s6
python -m pytest tests -m gpu -x -q > gpurun_out/s6/pytest.log 2>&1; tail -4 gpurun_out/s6/pytest.log | cut -c1-300
for sc in instances-10k; do
  for knob in "RTX_LEAF_MIN=1" "RTX_LEAF_MIN=8" "RTX_LEAF_MIN=20" "RTX_TRACE_GENERAL=big"; do
    env $knob python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s6/${sc}_$knob.json 2> gpurun_out/s6/${sc}_$knob.err
    python scripts/ab_line.py "$knob" $sc gpurun_out/s6/${sc}_$knob.json
  done
done
for knob in "RTX_LEAF_MIN=20" "RTX_LEAF_MIN=8" "RTX_LEAF_MIN=32" "RTX_LEAF_MIN_ANY=32"; do
    env $knob python bench.py --scene mis-spheres --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s6/ms_$knob.json 2> gpurun_out/s6/ms_$knob.err
    python scripts/ab_line.py "$knob" mis-spheres gpurun_out/s6/ms_$knob.json
done
for knob in "RTX_LEAF_MIN=16" "RTX_LEAF_MIN=20" "RTX_LEAF_MIN=24"; do
    env $knob python bench.py --scene blob --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s6/blob_$knob.json 2> gpurun_out/s6/blob_$knob.err
    python scripts/ab_line.py "$knob" blob gpurun_out/s6/blob_$knob.json
done
for knob in "RTX_LEAF_MIN_ANY=20" "RTX_LEAF_MIN_ANY=32" "RTX_LEAF_MIN_ANY=48"; do
    env $knob python bench.py --scene room --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s6/room_$knob.json 2> gpurun_out/s6/room_$knob.err
    python scripts/ab_line.py "$knob" room gpurun_out/s6/room_$knob.json
done
