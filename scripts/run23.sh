#!/bin/bash
mkdir -p gpurun_out/s23
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig.so
for rep in 1 2; do
for v in "a_lds16 0" "b_lds12 0" "b_lds12 4"; do
  set -- $v
  cp rustracer_amd/csrc/_build/ab/$1.so $LIB
  for sc in blob mis room; do
    RTX_TOP_BLOCKS_ANY=$2 timeout 300 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s23/$1_$2_${sc}_$rep.json 2> gpurun_out/s23/$1_$2_${sc}_$rep.err
    python scripts/ab_line.py "$1/any$2" $sc gpurun_out/s23/$1_$2_${sc}_$rep.json
  done
done
done
cp /tmp/orig.so $LIB
