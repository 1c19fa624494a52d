#!/bin/bash
mkdir -p gpurun_out/s37
for rep in 1 2 3 4; do
for q in 0 256 2304 4352 69888; do
    RTX_WS_STAGGER=$q timeout 300 python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s37/s${q}_$rep.json 2> gpurun_out/s37/s${q}_$rep.err
    python scripts/ab_line.py "stagger$q" cornell gpurun_out/s37/s${q}_$rep.json
done
done
