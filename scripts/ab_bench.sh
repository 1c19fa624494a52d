#!/bin/bash
# A/B on ONE box (box-to-box variation is +-2 %, and on one box back-to-back runs of one binary still differ by a few % in a stage): every librtx_hip.so
# variant under rustracer_amd/csrc/_build/ab/ is copied over the library in turn and benched on the scenes given (default: all four), 2 timed frames each;
# REPS=n repeats the whole round n times (variants interleaved: a b c a b c ...) and scripts/ab_summary.py prints the medians.
# Usage (GPU box, repo root): [REPS=3] [STEPS=2] bash scripts/ab_bench.sh [scene ...]
SCENES=${@:-cornell blob mis room}
REPS=${REPS:-1}
STEPS=${STEPS:-2}
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig_librtx_hip.so
mkdir -p gpurun_out/ab
for rep in $(seq 1 $REPS); do
for v in rustracer_amd/csrc/_build/ab/*.so; do
  name=$(basename $v .so)
  cp $v $LIB
  for sc in $SCENES; do
    python bench.py --scene $sc --steps $STEPS --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/${name}_${sc}_$rep.json > gpurun_out/ab/${name}_${sc}_$rep.line 2> gpurun_out/ab/${name}_${sc}_$rep.err
    python scripts/ab_line.py $name $sc gpurun_out/ab/${name}_${sc}_$rep.json
  done
done
done
cp /tmp/orig_librtx_hip.so $LIB
python scripts/ab_summary.py
