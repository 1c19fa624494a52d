#!/bin/bash
# A/B on ONE box (box-to-box variation is +-2 %): every librtx_hip.so variant under rustracer_amd/csrc/_build/ab/ is copied over the library in turn and
# benched on the scenes given (default: all four), 2 timed frames each. Usage (GPU box, repo root): bash scripts/ab_bench.sh [scene ...]
SCENES=${@:-cornell blob mis room}
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig_librtx_hip.so
mkdir -p gpurun_out/ab
for v in rustracer_amd/csrc/_build/ab/*.so; do
  name=$(basename $v .so)
  cp $v $LIB
  for sc in $SCENES; do
    python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/ab/${name}_${sc}.json 2> gpurun_out/ab/${name}_${sc}.err
    python scripts/ab_line.py $name $sc gpurun_out/ab/${name}_${sc}.json
  done
done
cp /tmp/orig_librtx_hip.so $LIB
